"""The N > 1 product path on the GPU box.  The box has ONE MI355X, so the ranks are fresh child
processes sharing GPU 0 (tests/dist_worker.py) and their accumulator blocks are summed over gloo on the host -- RCCL
does not form a communicator with two ranks on one device.  Everything else is what bench.py --gpus N and ResidentEm
run: utterances dealt with shard_utterances / take_utterances, K1 -> K2 -> K3 through the C-ABI per shard, K4 on the
summed block on every rank.  The RCCL entry points themselves (khg_comm_create, khg_accs_allreduce[_f32]) are covered
with a one-rank communicator, and bench.py's self-launch with KHG_BENCH_SHARE_GPU=1."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _spawn(mode, world, tmp_path):
    port = _port()
    outs = [str(tmp_path / f"{mode}_r{r}.npz") for r in range(world)]
    # every rank's stderr goes to its own file: a chatty rank can never fill a pipe while another rank waits for it in a collective
    logs = [open(tmp_path / f"{mode}_r{r}.err", "w+") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--out", outs[r]], cwd=ROOT, stderr=logs[r], text=True) for r in range(world)]
    try:
        for p in procs:
            p.wait(timeout=900)
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        raise
    finally:
        errs = []
        for f in logs:
            f.seek(0); errs.append(f.read()); f.close()
    for p, e in zip(procs, errs):
        assert p.returncode == 0, e[-3000:]
    return [np.load(o) for o in outs]


def test_two_ranks_sharded_em_pass_equals_single_process(ctx, tmp_path):
    import dist_worker as dw
    from kaldi_hmm_gmm_amd.dist import shard_utterances

    got = _spawn("kernels", 2, tmp_path)
    m, gc, cost, ut = dw.kernels_inputs()
    shards = shard_utterances(np.diff(ut.frame_off), 2)
    n_utt = len(ut.frame_off) - 1
    assert sorted(np.concatenate(shards).tolist()) == list(range(n_utt))
    PAR = ("gauss_off", "weights", "gconsts", "means_invvars", "inv_vars")

    # (1) every rank ends the pass with the SAME summed block and the SAME new model, bit for bit
    assert np.array_equal(got[0]["block"], got[1]["block"])
    for k in PAR:
        assert np.array_equal(got[0][k], got[1][k]), k
    assert int(got[0]["removed"]) == int(got[1]["removed"])
    # ... and the SHARDED M-step (each rank updates its own pdf range, the rows are exchanged, every rank finishes) gives that model too
    for r in range(2):
        for k in PAR:
            assert np.array_equal(got[r]["sharded_" + k], got[r][k]), (r, k)
        assert int(got[r]["sharded_removed"]) == int(got[r]["removed"]) and float(got[r]["sharded_objf"]) == float(got[r]["objf"])

    # (2) this process, running the two shards one after the other and adding the two blocks (a + b: the sum of two
    # operands does not depend on the order), reproduces the ranks' block, alignments and post-K4 model bit for bit
    blocks = []
    for r in range(2):
        dm, tm, us, accs, buf, ali = dw.kernels_pass(ctx, m, gc, cost, ut, shards[r])
        assert np.array_equal(got[r]["mine"], shards[r]) and np.array_equal(got[r]["ali"], ali)
        assert np.array_equal(got[r]["own_block"], buf)              # K1-K3 are run-to-run and process-to-process deterministic
        blocks.append(buf)
        if r == 0:
            for o in (accs, us, tm, dm):
                o.close()
    summed = blocks[0] + blocks[1]
    assert np.array_equal(summed, got[0]["block"])
    res, d = dw.kernels_mstep(dm, accs, summed)
    for k in PAR:
        assert np.array_equal(d[k], got[0][k]), k
    for o in (accs, us, tm, dm):
        o.close()

    # (3) against ONE process over the whole set: alignments identical per utterance, integer statistics exact, the
    # fp64 sums equal up to the association of the additions (1e-12), the new parameters to float rounding
    dm, tm, us, accs, whole, ali_all = dw.kernels_pass(ctx, m, gc, cost, ut, np.arange(n_utt))
    for r in range(2):
        fo = ut.frame_off
        want = np.concatenate([ali_all[fo[u]: fo[u + 1]] for u in shards[r]])
        assert np.array_equal(got[r]["ali"], want)
    sp = accs.split(whole)
    gp = accs.split(got[0]["block"].copy())
    assert np.array_equal(sp["trans_acc"], gp["trans_acc"]) and sp["total_frames"] == gp["total_frames"] == ut.frame_off[-1]
    np.testing.assert_allclose(gp["occ"], sp["occ"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gp["mean_acc"], sp["mean_acc"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(gp["var_acc"], sp["var_acc"], rtol=1e-12, atol=1e-9)
    assert gp["total_log_like"] == pytest.approx(sp["total_log_like"], rel=1e-12)
    res1, d1 = dw.kernels_mstep(dm, accs, whole)
    assert np.array_equal(d1["gauss_off"], got[0]["gauss_off"]) and res1["removed"] == int(got[0]["removed"])
    np.testing.assert_allclose(got[0]["weights"], d1["weights"], rtol=3e-7)
    np.testing.assert_allclose(got[0]["inv_vars"], d1["inv_vars"], rtol=2e-6)
    np.testing.assert_allclose(got[0]["means_invvars"], d1["means_invvars"], rtol=2e-6, atol=1e-6)
    for o in (accs, us, tm, dm):
        o.close()


def test_resident_em_two_ranks_hold_identical_models(ctx, tmp_path):
    """every rank must split (mix up) the same way.  Two ranks run ResidentEm on their shards for four
    passes with mixing up and the default perturbation stream: bit-identical models on both ranks, the same number of
    Gaussians and (to rounding) the same likelihoods as one process over the whole set."""
    import dist_worker as dw
    import kaldi_hmm_gmm_amd as khg

    got = _spawn("resident", 2, tmp_path)
    for k in ("gauss_off", "weights", "gconsts", "means_invvars", "inv_vars", "log_probs", "log"):
        assert np.array_equal(got[0][k], got[1][k]), k
    ex, tm, am, graphs, feats, ali = dw.resident_inputs()
    em = khg.ResidentEm(am, tm, graphs, feats, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1, ctx=ctx)
    em.set_alignments(ali)
    log = np.asarray(dw.resident_run(em, ex), np.float64)
    go = em.sync_host().flat()[0]
    em.close()
    assert np.array_equal(go, got[0]["gauss_off"]) and go[-1] > 11            # mixed up, and identically so
    assert np.array_equal(log[:, 0], got[0]["log"][:, 0]) and log[0, 0] == sum(f.shape[0] for f in feats)   # frames: all of them, every pass
    np.testing.assert_allclose(got[0]["log"][:, 1], log[:, 1], rtol=1e-5)


def test_rccl_entry_points_with_a_one_rank_communicator(ctx):
    """khg_comm_unique_id / khg_comm_create / khg_accs_allreduce / khg_accs_allreduce_f32 on the real RCCL (bound at run
    time from the copy torch loaded): the sum over one rank is the identity; the fp32 wire rounds the block to float."""
    import dist_worker as dw
    from kaldi_hmm_gmm_amd import Comm

    m, gc, cost, ut = dw.kernels_inputs()
    dm, tm, us, accs, buf, _ = dw.kernels_pass(ctx, m, gc, cost, ut, np.arange(8))
    comm = Comm(ctx, 1, 0, Comm.unique_id())
    accs.allreduce(comm)
    ctx.sync()
    assert np.array_equal(accs.download_range(0, accs.size), buf)
    accs.allreduce(comm, wire_fp32=True)
    ctx.sync()
    assert np.array_equal(accs.download_range(0, accs.size), buf.astype(np.float32).astype(np.float64))
    accs.upload(buf)
    accs.allreduce(None)                                   # one-rank job: no communicator, nothing to do
    accs.allreduce(None, wire_fp32=True)                   # only the rounding
    ctx.sync()
    assert np.array_equal(accs.download_range(0, accs.size), buf.astype(np.float32).astype(np.float64))
    # C1 pipelined behind K3 by pdf ranges (khg_acc_stats_reduce) and the range exchange on its own (khg_accs_allreduce_range):
    # with one rank every sum is the identity, so the block must come out bit for bit as khg_acc_stats left it -- for the
    # wave-local K3 and for the block / VALU forms, whose range launches take the pdf offset too
    for form in (0, 1, 2):
        old = ctx.set_option("k3_form", form)
        try:
            for nparts in (1, 3, 4, 1000):
                accs.zero()
                us.acc_stats_reduce(dm, tm, accs, 1.0, comm, nparts)
                ctx.sync()
                got = accs.download_range(0, accs.size)
                accs.zero()
                us.acc_stats(dm, tm, accs)
                ctx.sync()
                assert np.array_equal(got, accs.download_range(0, accs.size)), (form, nparts)
        finally:
            ctx.set_option("k3_form", old)
    accs.zero()
    us.acc_stats(dm, tm, accs)
    ctx.sync()
    assert np.array_equal(accs.download_range(0, accs.size), buf)
    P = len(m.gauss_off) - 1
    for p0, n in ((0, P // 3), (P // 3, P - P // 3), (-1, 0)):
        accs.allreduce_range(dm, p0, n, comm)
    ctx.sync()
    assert np.array_equal(accs.download_range(0, accs.size), buf)
    with pytest.raises(Exception):
        accs.allreduce_range(dm, P - 1, 5, comm)
    # the sharded M-step through RCCL with one rank (ncclReduce / ncclBroadcast to itself) == the replicated one, Gaussians removed included
    from kaldi_hmm_gmm_amd import DeviceModel
    from kaldi_hmm_gmm_amd.mle import MleDiagGmmOptions
    opts = MleDiagGmmOptions(min_gaussian_occupancy=40.0)
    dm_a = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    dm_b = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    accs.upload(buf)
    ra = dm_a.mle_update(accs, opts, 0x7)
    accs.upload(buf)
    rb = dm_b.mle_update_sharded(accs, opts, 0x7, comm)
    assert ra == rb and ra["removed"] > 0
    da, db = dm_a.download(), dm_b.download()
    for k in da:
        assert np.array_equal(da[k], db[k]), k
    # two "ranks" emulated in this process: two copies of the model each update one half, swap rows, finish
    dm_c = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    dm_d = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    accs.upload(buf)
    h = P // 2 + 1
    dm_c.mle_update_range(accs, opts, 0x7, 0, h)
    dm_d.mle_update_range(accs, opts, 0x7, h, P - h)
    dm_d.mle_rows_upload(dm_c.mle_rows_download(0, h))
    dm_c.mle_rows_upload(dm_d.mle_rows_download(h, P - h))
    rc_, rd_ = dm_c.mle_update_finish(), dm_d.mle_update_finish()
    assert rc_ == ra and rd_ == ra
    for dmx in (dm_c, dm_d):
        dx = dmx.download()
        for k in da:
            assert np.array_equal(da[k], dx[k]), k
    for o in (dm_a, dm_b, dm_c, dm_d):
        o.close()
    comm.close()
    for o in (accs, us, tm, dm):
        o.close()


def test_bench_launches_its_own_ranks_and_shards_the_one_set():
    """`python bench.py --gpus 2` from a bare shell (no torchrun around it): the parent starts the ranks, relays ONE
    JSON line; the two ranks own the frames of the N = 1 set (same seeds) and the summed accumulators count all of
    them.  KHG_BENCH_SHARE_GPU=1: both ranks on GPU 0, block summed on the host."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--utts", "3000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r1 = subprocess.run(base, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads(r1.stdout.strip().splitlines()[-1])
    r2 = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=900, env=dict(env, KHG_BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-3000:]
    lines = r2.stdout.strip().splitlines()
    assert len(lines) == 1, r2.stdout[:2000]
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1 and d2["allreduce"] == "host"
    f = d1["config"]["frames_per_step"]
    assert d2["config"]["frames_per_step"] == f == d1["check"]["frames_in_set"] == d2["check"]["frames_in_set"]
    assert d1["check"]["acc_total_frames"] == f and d2["check"]["acc_total_frames"] == f        # rank 0 holds the sum over both shards
    assert d2["check"]["avg_loglike_per_frame"] == pytest.approx(d1["check"]["avg_loglike_per_frame"], rel=1e-9)
    assert d2["allreduce_ms_per_step"] is not None and d2["allreduce_bytes"] > 0 and d2["value"] > 0
    assert d2["roofline"]["frac_executed"] <= d2["roofline"]["frac"] < 1.0


def _check_rccl_ranks(got, world):
    PAR = ("gauss_off", "weights", "gconsts", "means_invvars", "inv_vars")
    tot = sum(g["own_block"] for g in got)
    scale = np.abs(tot).max()
    for r in range(world):
        # the three forms of C1 give the sum of the ranks' blocks: the host (gloo) sum, one ncclAllReduce, the pipelined pieces
        np.testing.assert_allclose(got[r]["host_block"], tot, rtol=1e-12, atol=1e-12 * scale)
        np.testing.assert_allclose(got[r]["whole_block"], tot, rtol=1e-12, atol=1e-12 * scale)
        np.testing.assert_allclose(got[r]["piped_block"], tot, rtol=1e-12, atol=1e-12 * scale)
        assert np.array_equal(got[r]["whole_block"], got[0]["whole_block"])        # every rank holds the same bits
        assert np.array_equal(got[r]["piped_block"], got[0]["piped_block"])
        # the sharded M-step over the communicator = the replicated one on the summed block
        assert int(got[r]["sharded_removed"]) == int(got[r]["removed"])
        for k in PAR[:1]:
            assert np.array_equal(got[r]["sharded_" + k], got[r][k]), (r, k)
        for k in PAR[1:]:
            np.testing.assert_allclose(got[r]["sharded_" + k], got[r][k], rtol=2e-6, atol=1e-6, err_msg=k)
            assert np.array_equal(got[r]["sharded_" + k], got[0]["sharded_" + k]), (r, k)


def test_rccl_worker_with_one_rank(tmp_path):
    """tests/dist_worker.py rccl on the one GPU of this box: a one-rank communicator through every RCCL entry point the
    multi-GPU run uses (ncclAllReduce whole and in pdf-range groups behind K3, ncclReduce / ncclBroadcast of the sharded M-step)."""
    _check_rccl_ranks(_spawn("rccl", 1, tmp_path), 1)


def test_rccl_two_ranks_on_two_gpus(tmp_path):
    """The product's own exchange with N > 1: one process per GPU, RCCL over xGMI.  Needs two devices: skipped on the one-GPU
    test box (the only place this path can run is a multi-GPU node)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (RCCL does not form a communicator with two ranks on one device)")
    world = 2 if n < 4 else 4
    _check_rccl_ranks(_spawn("rccl", world, tmp_path), world)


@pytest.mark.parametrize("nparts", [1, 4])
def test_split_mode_and_mixed_classes_with_the_exchange_behind_k3(ctx, nparts):
    """khg_acc_stats_reduce on a real (one-rank) RCCL communicator in the two situations where K3 is more than one launch sequence:
    the SPLIT mode (order-faithful decoders still running: certified utterances first, the rest in a second pass -- the exchange's
    pieces follow the SECOND pass) and a model with pdfs of two classes (wave form + chunk-per-block form: the pieces follow the
    second class).  With one rank every sum is the identity: the block must equal the one khg_acc_stats leaves without an exchange."""
    from kaldi_hmm_gmm_amd import Comm, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth
    from oracle import oracle as orc
    P, D, U = 90, 40, 200
    counts = np.full(P, 64); counts[7] = 130; counts[40] = 100
    m0 = synth.make_model(P, 64, D, seed=77, gauss_counts=counts)
    ut = synth.make_utts(m0, U, seed=8, min_phones=4, max_phones=12)
    m = m0
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    il = np.arange(m.num_tids + 1, dtype=np.int32)
    cost = orc.add_transition_probs(il, np.zeros(m.num_tids + 1, np.float32), m.log_probs, m.non_self_loop_log_probs, m.id2state, m.is_self_loop, 1.0, 0.1)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    poff, pdfs = us.pdf_lists()
    rng = np.random.default_rng(4)
    # scores that make a narrow beam prune: a good share of the utterances leaves the beam certificate
    us.upload_loglikes([(-40.0 * rng.random((poff[u + 1] - poff[u], int(ut.frame_off[u + 1] - ut.frame_off[u])))).astype(np.float32) for u in range(U)])
    comm = Comm(ctx, 1, 0, Comm.unique_id())
    res = us.align(tm, beam=3.0, retry_beam=30.0, acoustic_scale=0.1)
    n_fb = int((res["status"] & 8 != 0).sum())
    assert 10 <= n_fb, n_fb
    want = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, want)                               # synchronous alignment above: one pass, no exchange
    w = want.download()
    for split in (0, 1):
        old = ctx.set_option("k2_split", split)
        try:
            us.align(tm, beam=3.0, retry_beam=30.0, acoustic_scale=0.1, download=False)
            got = DeviceAccs(ctx, dm, tm)
            us.acc_stats_reduce(dm, tm, got, 1.0, comm, nparts)
            g = got.download()
        finally:
            ctx.set_option("k2_split", old)
        assert np.array_equal(np.asarray(us.download_ali()), res["ali"])
        assert np.array_equal(g["trans_acc"], w["trans_acc"]) and g["total_frames"] == w["total_frames"]
        for k in ("occ", "mean_acc", "var_acc"):
            np.testing.assert_allclose(g[k], w[k], rtol=1e-6, atol=1e-6 * np.abs(w[k]).max(), err_msg=f"{k} split {split} nparts {nparts}")
        got.close()
    want.close(); comm.close(); us.close(); tm.close(); dm.close()
