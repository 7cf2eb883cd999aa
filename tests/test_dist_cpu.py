"""N > 1 path on CPU: two gloo ranks shard the utterances, accumulate their shards with the oracle,
all-reduce the fp64 accumulator block and must reproduce the single-process sums -- the same
shard/all-reduce code bench.py and the library use with RCCL on the GPU box."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kaldi_hmm_gmm_amd import synth
    from kaldi_hmm_gmm_amd.dist import allreduce_accs, shard_utterances
    from oracle import oracle as orc

    m = synth.make_model(12, 3, 6, seed=4)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    om = orc.OModel(m.gauss_off, gc, m.means_invvars, m.inv_vars)
    ut = synth.make_utts(m, 9, seed=9, min_phones=2, max_phones=4)
    mine = shard_utterances(np.diff(ut.frame_off), world)[rank]
    acc = orc.OAccs(int(m.gauss_off[-1]), m.dim, m.num_tids)
    for u in mine:
        sl = slice(ut.frame_off[u], ut.frame_off[u + 1])
        orc.acc_stats_ali(om, m.id2pdf, ut.feats[sl], ut.ref_ali[sl], acc)
    block = torch.from_numpy(np.concatenate([acc.occ, acc.mean_acc.ravel(), acc.var_acc.ravel(), acc.trans_acc,
                                             [acc.total_frames, acc.total_log_like]]))
    allreduce_accs(block)
    if rank == 0:
        np.save(out, block.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from kaldi_hmm_gmm_amd import synth
    from oracle import oracle as orc

    out = str(tmp_path / "acc.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    m = synth.make_model(12, 3, 6, seed=4)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    om = orc.OModel(m.gauss_off, gc, m.means_invvars, m.inv_vars)
    ut = synth.make_utts(m, 9, seed=9, min_phones=2, max_phones=4)
    acc = orc.OAccs(int(m.gauss_off[-1]), m.dim, m.num_tids)
    for u in range(9):
        sl = slice(ut.frame_off[u], ut.frame_off[u + 1])
        orc.acc_stats_ali(om, m.id2pdf, ut.feats[sl], ut.ref_ali[sl], acc)
    want = np.concatenate([acc.occ, acc.mean_acc.ravel(), acc.var_acc.ravel(), acc.trans_acc,
                           [acc.total_frames, acc.total_log_like]])
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-12)
    assert got[-2] == ut.frame_off[-1]


def test_take_utterances_is_the_per_utterance_slice():
    """dist.take_utterances (what bench.py and the N > 1 tests cut a rank's shard with): the sub-set's CSR graphs, frame
    offsets and global frame indices equal the per-utterance slices of the whole set."""
    sys.path.insert(0, ROOT)
    from kaldi_hmm_gmm_amd import synth
    from kaldi_hmm_gmm_amd.dist import shard_utterances, take_utterances

    m = synth.make_model(30, 2, 5, seed=3)
    ut = synth.make_utts(m, 23, seed=12, min_phones=2, max_phones=5, feats=False)
    shards = shard_utterances(np.diff(ut.frame_off), 3)
    assert sorted(np.concatenate(shards).tolist()) == list(range(23))
    loads = [int(np.diff(ut.frame_off)[s].sum()) for s in shards]
    assert max(loads) - min(loads) <= int(np.diff(ut.frame_off).max())
    g = ut.graphs
    for idx in shards + [np.arange(23), np.array([22]), np.array([0, 7])]:
        fo, gl, fr = take_utterances(ut.frame_off, g, idx)
        assert fo[0] == 0 and gl["state_off"][0] == 0 and gl["arc_off"][0] == 0
        for k, u in enumerate(idx):
            T = ut.frame_off[u + 1] - ut.frame_off[u]
            assert fo[k + 1] - fo[k] == T
            assert np.array_equal(fr[fo[k]: fo[k + 1]], np.arange(ut.frame_off[u], ut.frame_off[u + 1]))
            s0, s1 = g["state_off"][u], g["state_off"][u + 1]
            l0, l1 = gl["state_off"][k], gl["state_off"][k + 1]
            assert l1 - l0 == s1 - s0 and gl["start"][k] == g["start"][u]
            assert np.array_equal(gl["final"][l0:l1], g["final"][s0:s1])
            a0, a1 = g["arc_off"][s0], g["arc_off"][s1]
            b0, b1 = gl["arc_off"][l0], gl["arc_off"][l1]
            assert np.array_equal(gl["arc_off"][l0: l1 + 1] - b0, g["arc_off"][s0: s1 + 1] - a0)
            for key in ("ilabel", "olabel", "weight", "nextstate"):
                assert np.array_equal(gl[key][b0:b1], g[key][a0:a1]), key
