#!/usr/bin/env python3
"""One rank of the N > 1 product path, run as a FRESH child process by tests/test_gpu_dist.py (several ranks share GPU 0
on the one-GPU test box; RCCL refuses two ranks on one device, so the accumulator blocks are summed over gloo on the
host -- everything else is the product path: shard_utterances / take_utterances, K1 -> K2 -> K3 through the C-ABI on the
rank's shard, K4 on the summed block).

  kernels  : synthetic model + set; saves the summed block, the post-K4 parameters and the shard's alignments
  resident : ResidentEm over a shard of the YES/NO recipe data, several passes with mixing up; saves the final model
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

KERNELS_SHAPE = dict(num_pdfs=90, gauss=24, dim=40, n_utt=48, model_seed=31, utt_seed=5)


def kernels_inputs():
    from kaldi_hmm_gmm_amd import _lib, synth
    import ctypes as C

    k = KERNELS_SHAPE
    m = synth.make_model(k["num_pdfs"], k["gauss"], k["dim"], seed=k["model_seed"])
    gc = np.zeros(m.weights.shape[0], np.float32)
    _lib.check(_lib.lib.khg_compute_gconsts(m.num_pdfs, m.dim, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float),
                                            _lib.ptr(m.inv_vars, C.c_float), _lib.ptr(m.means_invvars, C.c_float),
                                            _lib.ptr(gc, C.c_float), None))
    cost = np.zeros(m.num_tids + 1, np.float32)
    _lib.check(_lib.lib.khg_scaled_trans_cost(m.num_tids, _lib.ptr(m.log_probs, C.c_float), _lib.ptr(m.non_self_loop_log_probs, C.c_float),
                                              _lib.ptr(m.id2state, C.c_int32), _lib.ptr(m.is_self_loop, C.c_uint8), 1.0, 0.1,
                                              _lib.ptr(cost, C.c_float)))
    ut = synth.make_utts(m, k["n_utt"], seed=k["utt_seed"], min_phones=3, max_phones=8)
    return m, gc, cost, ut


def kernels_pass(ctx, m, gc, cost, ut, idx):
    """K1 (reachable cells) -> K2 -> K3 on the utterances `idx` of the set: -> (block, alignment of those utterances)."""
    from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet
    from kaldi_hmm_gmm_amd.dist import take_utterances

    fo, g, fr = take_utterances(ut.frame_off, ut.graphs, idx)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, fo, ut.feats[fr], graphs=g)
    us.loglikes(dm, reachable_only=True)
    r = us.align(tm, acoustic_scale=0.1)
    assert ((r["status"] & 1) == 0).all()
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs)
    buf = np.zeros(accs.size, np.float64)
    from kaldi_hmm_gmm_amd import _lib
    import ctypes as C
    _lib.check(_lib.lib.khg_accs_download(ctx.h, accs.h, _lib.ptr(buf, C.c_double)))
    return dm, tm, us, accs, buf, r["ali"]


def kernels_mstep(dm, accs, block):
    accs.upload(block)
    r = dm.mle_update(accs, None, 0x7)
    d = dm.download()
    return r, d


def run_kernels(args):
    import torch
    import torch.distributed as dist

    from kaldi_hmm_gmm_amd import Context
    from kaldi_hmm_gmm_amd.dist import shard_utterances

    m, gc, cost, ut = kernels_inputs()
    mine = shard_utterances(np.diff(ut.frame_off), args.world)[args.rank]
    ctx = Context(0)
    dm, tm, us, accs, buf, ali = kernels_pass(ctx, m, gc, cost, ut, mine)
    own = buf.copy()
    t = torch.from_numpy(buf)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)              # C1 on the host (see the module docstring)
    # the SHARDED M-step (khg_model_mle_update_range / _rows_* / _finish) on a second copy of the model: this rank
    # updates its own pdf range, the rows travel over the process group (the stand-in for ncclBroadcast on one GPU), every rank
    # finishes -- must equal the replicated update below bit for bit
    from kaldi_hmm_gmm_amd import DeviceModel
    dm2 = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    accs.upload(buf)
    P = len(m.gauss_off) - 1
    rng_of = lambda r: (P * r // args.world, P * (r + 1) // args.world - P * r // args.world)
    p0, npd = rng_of(args.rank)
    dm2.mle_update_range(accs, None, 0x7, p0, npd)
    rows = [None] * args.world
    dist.all_gather_object(rows, {k: (np.asarray(v) if not isinstance(v, int) else v) for k, v in dm2.mle_rows_download(p0, npd).items()})
    for r_, rw in enumerate(rows):
        if r_ != args.rank:
            dm2.mle_rows_upload(rw)
    r2 = dm2.mle_update_finish()
    d2 = dm2.download()
    r, d = kernels_mstep(dm, accs, buf)
    np.savez(args.out, own_block=own, block=buf, ali=ali, mine=mine, removed=r["removed"], objf=r["objf_change"],
             sharded_removed=r2["removed"], sharded_objf=r2["objf_change"], **{"sharded_" + k: v for k, v in d2.items()},
             **{k: v for k, v in d.items()})


def run_rccl(args):
    """One rank per GPU (device = rank): the product's own exchange.  K1 -> K2 -> K3 on the shard, then C1 three ways on copies of
    the rank's block -- khg_accs_allreduce (one ncclAllReduce), khg_acc_stats_reduce (pipelined by pdf ranges behind K3) and the
    gloo sum on the host as the yardstick -- and the sharded M-step over the communicator (ncclReduce / ncclBroadcast)."""
    import torch
    import torch.distributed as dist

    from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, _lib
    from kaldi_hmm_gmm_amd.dist import make_comm, shard_utterances
    import ctypes as C

    m, gc, cost, ut = kernels_inputs()
    mine = shard_utterances(np.diff(ut.frame_off), args.world)[args.rank]
    ndev = torch.cuda.device_count()
    ctx = Context(args.rank % max(ndev, 1))
    comm = make_comm(ctx)                                   # None in a one-rank group
    if comm is None and args.world == 1:
        from kaldi_hmm_gmm_amd import Comm
        comm = Comm(ctx, 1, 0, Comm.unique_id())
    dm, tm, us, accs, buf, ali = kernels_pass(ctx, m, gc, cost, ut, mine)
    own = buf.copy()
    host = buf.copy()
    dist.all_reduce(torch.from_numpy(host), op=dist.ReduceOp.SUM)      # yardstick: the gloo sum on the host

    def download(a):
        b = np.zeros(a.size, np.float64)
        _lib.check(_lib.lib.khg_accs_download(ctx.h, a.h, _lib.ptr(b, C.c_double)))
        return b

    accs.allreduce(comm)                                    # (1) one all-reduce of the whole block
    whole = download(accs)
    accs2 = DeviceAccs(ctx, dm, tm)                         # (2) K3 again with C1 pipelined behind it by pdf ranges
    us.acc_stats_reduce(dm, tm, accs2, 1.0, comm, 4)
    piped = download(accs2)
    accs3 = DeviceAccs(ctx, dm, tm)                         # (3) the sharded M-step from the rank's OWN (unreduced) block
    accs3.upload(own)
    dm3 = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    r3 = dm3.mle_update_sharded(accs3, None, 0x7, comm)
    d3 = dm3.download()
    r, d = kernels_mstep(dm, accs, host)                    # the replicated update from the host-summed block
    np.savez(args.out, own_block=own, host_block=host, whole_block=whole, piped_block=piped, ali=ali, mine=mine, removed=r["removed"],
             objf=r["objf_change"], sharded_removed=r3["removed"], sharded_objf=r3["objf_change"],
             **{"sharded_" + k: v for k, v in d3.items()}, **{k: v for k, v in d.items()})
    comm.close()


def resident_inputs():
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd.training_graph import TrainingGraphCompiler, equal_align, generate_hmm_topo
    import train_mono_synthetic as ex

    rng = np.random.default_rng(17)
    utts = ex.make_data(24, 13, rng)
    feats = [u[2] for u in utts]
    topo = generate_hmm_topo(non_sil_phones=[ex.Y, ex.N], sil_phone=ex.SIL)
    tm, tree, am = khg.gmm_init_mono(topo, np.concatenate(feats[:10]))
    comp = TrainingGraphCompiler(tm, tree, {ex.YES: [(1.0, [ex.Y])], ex.NO: [(1.0, [ex.N])]}, sil_phone=ex.SIL, sil_prob=0.5)
    graphs = comp.compile_graphs_from_text([u[1] for u in utts])
    ali = []
    for g, x in zip(graphs, feats):
        ok, a = equal_align(g, x.shape[0], rand_seed=3, num_retries=10)
        assert ok
        ali.append(a)
    return ex, tm, am, graphs, feats, ali


def resident_run(em, ex, n_pass=4):
    """Passes of the recipe with mixing up; split perturbations come from ResidentEm's own default stream."""
    import kaldi_hmm_gmm_amd as khg

    cfg = khg.AlignConfig(beam=6.0, retry_beam=40.0, careful=False)
    tcfg = khg.MleTransitionUpdateConfig()
    opts = khg.MleDiagGmmOptions(min_gaussian_occupancy=3)
    log = []
    for it, target in enumerate([11, 15, 20, 22][:n_pass]):
        if it > 0:
            em.boost_silence([ex.SIL], boost=1.25)
            em.align(cfg)
        st = em.accumulate()
        info = em.update(tcfg, opts, mixup=target, update_flags="mvwt")
        log.append((st["total_frames"], info["avg_like"], em.num_gauss))
    return log


def run_resident(args):
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd import Context
    from kaldi_hmm_gmm_amd.dist import shard_utterances

    ex, tm, am, graphs, feats, ali = resident_inputs()
    mine = shard_utterances([f.shape[0] for f in feats], args.world)[args.rank]
    ctx = Context(0)
    em = khg.ResidentEm(am, tm, [graphs[i] for i in mine], [feats[i] for i in mine], acoustic_scale=0.1, transition_scale=1.0,
                        self_loop_scale=0.1, ctx=ctx)
    em.set_alignments([ali[i] for i in mine])
    log = resident_run(em, ex)
    go, gc, w, miv, iv = em.sync_host().flat()
    lp = np.asarray([tm.get_transition_log_prob(t) for t in range(1, tm.num_transition_ids + 1)], np.float32)
    np.savez(args.out, gauss_off=go, gconsts=gc, weights=w, means_invvars=miv, inv_vars=iv, log_probs=lp, log=np.asarray(log, np.float64))
    em.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["kernels", "resident", "rccl"])
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port))
    dist.init_process_group("gloo", rank=args.rank, world_size=args.world)
    try:
        {"kernels": run_kernels, "resident": run_resident, "rccl": run_rccl}[args.mode](args)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
