#!/usr/bin/env python3
"""bench.py -- frames/sec (whole node) for one EM iteration = align (K1 log-likes + K2 Viterbi)
+ acc-stats (K3) [+ RCCL all-reduce of the accumulators when N > 1] on BASELINE.json's
5000-pdf x 64-Gaussian, 40-dim synthetic workload (configs[3]).

Contract: `python bench.py --gpus N --steps K --warmup W`; one rank per GPU (RANK/LOCAL_RANK/
WORLD_SIZE from the env under torch.distributed.run); rank 0 prints ONE JSON line.  Run from a bare shell
with --gpus N > 1 it starts the N ranks itself (fresh child processes through torch.distributed.run, before
anything in this process has touched a GPU) and relays rank 0's line.
Strong scaling as configs[3] states it: the ONE 100k-utterance set (the N = 1 workload, same seeds, same
frames) is dealt to the N ranks by total frames (kaldi_hmm_gmm_amd.dist.shard_utterances).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_lines  # noqa: E402  (the side lines, the launch ladder, the CPU baseline and the oracle check)
from bench_lines import check_vs_oracle, cpu_baseline, csrc_sha, per_call_line, pmc_traffic, self_launch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense" (the bare v_mfma_f32_32x32x16_bf16 loop of
#                                 tools/mfma_bf16_chain.hip holds 1.78-1.85 PF on this chip: the clock drops to ~1.8 GHz under it)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="tri5000x64")
    ap.add_argument("--utts", type=int, default=100000, help="total utterances over all ranks")
    ap.add_argument("--beam", type=float, default=200.0)
    ap.add_argument("--retry-beam", type=float, default=0.0)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batches", type=int, default=1, help="utterance batches per rank (more batches bound the score/back-pointer buffers and let a batch's serial fallback decoder overlap the next batch's K1)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the batches alternate over")
    ap.add_argument("--full-loglikes", action="store_true", help="K1 over every (frame, pdf) cell instead of only those a decoder token can read")
    ap.add_argument("--no-band", action="store_true", help="K1 from every pdf's first readable tile to the utterance's end (khg_loglikes_reachable) instead of the band "
                    "[first readable, last useful] tile of khg_loglikes_band (the default at beam >= 100, f16x2s form)")
    ap.add_argument("--k1", choices=["auto", "f16x2s", "f16x2", "pdf", "utt"], default="auto",
                    help="K1 arithmetic / tiling (khg_ctx_set_k1_form): auto = f16x2s (fp16 matrix cores at fp32 accuracy, 3 partial "
                         "products); pdf / utt = the fp32-MFMA forms")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the two extra steps that time the fp32-MFMA K1 beside an f16x2s / f16x2 run")
    ap.add_argument("--seed", type=int, default=20230418)
    ap.add_argument("--transcripts", choices=["uniform", "zipf"], default="uniform",
                    help="phone sequences: independent uniform phones (BASELINE.json's synthetic set) or running text from a Zipf lexicon "
                         "(utterances share pdfs the way real transcripts do: synth.zipf_phone_stream)")
    ap.add_argument("--allreduce", choices=["khg", "torch", "khg-f32", "host"], default="khg",
                    help="C1: khg = khg_accs_allreduce (RCCL called by the library on the kernels' stream); torch = "
                         "torch.distributed.all_reduce on a view of the block; khg-f32 = the fp32-wire tolerance experiment; "
                         "host = block summed over gloo on the host (test rig: KHG_BENCH_SHARE_GPU=1 puts every rank on GPU 0, "
                         "where RCCL refuses to form a communicator)")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="form the process group and the library's RCCL communicator, all-reduce 1 MB through both, print one JSON line and "
                         "exit -- seconds, before any data is built: what the self-launcher runs first so that a broken RCCL costs no synthesis")
    ap.add_argument("--per-call-utts", type=int, default=int(os.environ.get("KHG_BENCH_PERCALL_UTTS", "256")),
                    help="utterances pushed through gmm_align_compiled + gmm_acc_stats_ali ONE CALL EACH after the timed region (per_call_line); 0 = skip")
    ap.add_argument("--no-recipe-beam-line", action="store_true", help="skip the two extra steps at the recipe's beam 6 / retry 40 (recipe_beam_line)")
    ap.add_argument("--flat-fraction", type=float, default=0.1,
                    help="flat_start_line: share of the pdfs that trade parameters in the mismatched scoring model (synth.mismatched_model)")
    ap.add_argument("--c1-parts", type=int, default=4,
                    help="--allreduce khg: C1 pipelined behind K3 in this many pdf ranges (khg_acc_stats_reduce); 1 = one all-reduce of the "
                         "whole block behind K3 (khg_accs_allreduce)")
    return ap.parse_args()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # stdout carries exactly ONE line, the JSON record of rank 0.  RCCL prints a version banner through C stdio on
    # stdout (flushed at exit, i.e. AFTER anything Python printed), torch may warn there too: everything else in
    # the process is pointed at stderr and the record is written to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    asked_allreduce = args.allreduce
    share_gpu = os.environ.get("KHG_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local, args.allreduce = 0, "host"
    backend = "gloo" if args.allreduce == "host" else "nccl"
    torch.cuda.set_device(local)
    # KHG_BENCH_FORCE_DIST=1: run the collective code path (process group, all-reduce of the accumulator block, max /
    # sum of the timings) in a one-rank group -- the only way to exercise it on a one-GPU box
    dist_on = world > 1 or os.environ.get("KHG_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        if world > 1:
            dist.init_process_group(backend, **kw)
        else:
            dist.init_process_group(backend, rank=0, world_size=1, **kw)

    from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth
    from kaldi_hmm_gmm_amd import _lib
    import ctypes as C

    # test hook of the launch ladder (tests/test_bench_contract.py): the exchange as asked for on the command line "cannot form its
    # communicator" -- every rank exits 3 the way a real failure does, and the self-launcher has to get a line out of a later rung
    asked_khg_piped = asked_allreduce == "khg" and args.c1_parts > 1
    if os.environ.get("KHG_BENCH_FAIL_COMM") == "1" and dist_on and asked_khg_piped:
        print(f"bench.py: rank {rank}/{world}: KHG_BENCH_FAIL_COMM=1: pretending the library's RCCL communicator could not be formed", file=sys.stderr, flush=True)
        os._exit(3)
    if args.dist_selftest:
        # seconds, before any data: process group -> (library communicator) -> a 1 MB accumulator block all-reduced the way the bench
        # is about to do it; the sums are checked
        from kaldi_hmm_gmm_amd.dist import make_comm
        t_s = time.perf_counter()
        sm = synth.make_model(25, 64, 40, seed=1)                    # 1600 Gaussians x 81 doubles = 1.04 MB of accumulators
        sgc = np.zeros(sm.weights.shape[0], np.float32)
        _lib.check(_lib.lib.khg_compute_gconsts(25, 40, _lib.ptr(sm.gauss_off, C.c_int32), _lib.ptr(sm.weights, C.c_float), _lib.ptr(sm.inv_vars, C.c_float),
                                                _lib.ptr(sm.means_invvars, C.c_float), _lib.ptr(sgc, C.c_float), None))
        sctx = Context(local)
        sdm = DeviceModel(sctx, sm.gauss_off, sgc, sm.means_invvars, sm.inv_vars)
        stm = DeviceTransitions(sctx, sm.id2pdf)
        sacc = DeviceAccs(sctx, sdm, stm)
        blk = np.full(sacc.size, float(rank + 1), np.float64)
        sacc.upload(blk)
        how = args.allreduce
        if dist_on and how in ("khg", "khg-f32"):
            scomm = make_comm(sctx, one_rank=True, timeout_s=float(os.environ.get("KHG_BENCH_COMM_TIMEOUT", "180")))
            sacc.allreduce(scomm)
            sctx.sync()
            got = sacc.download()["occ"][0]
            scomm.close()
        elif dist_on and how == "torch":
            tt = sacc.as_torch()
            dist.all_reduce(tt)
            torch.cuda.synchronize()
            got = float(tt[0])
        elif dist_on:
            tt = torch.from_numpy(blk)
            dist.all_reduce(tt)
            got = float(tt[0])
        else:
            got = 1.0
        want = world * (world + 1) / 2.0
        ok = abs(got - want) < 1e-9
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            os.write(json_fd, (json.dumps({"selftest": "ok" if ok else "WRONG SUM", "allreduce": how, "n_gpus": world, "sum": got, "want": want,
                                           "bytes": int(sacc.size) * 8, "seconds": time.perf_counter() - t_s}) + "\n").encode())
        os.close(json_fd)
        sys.exit(0 if ok else 4)

    P, G, D = synth.CONFIGS[args.config]
    model = synth.make_model(P, G, D, seed=args.seed)
    gc = np.zeros(model.weights.shape[0], np.float32)
    _lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(model.gauss_off, C.c_int32), _lib.ptr(model.weights, C.c_float),
                                            _lib.ptr(model.inv_vars, C.c_float), _lib.ptr(model.means_invvars, C.c_float),
                                            _lib.ptr(gc, C.c_float), None))
    # AddTransitionProbs scales of the yesno recipe (egs/yesno/train.py:179-181)
    cost = np.zeros(model.num_tids + 1, np.float32)
    _lib.check(_lib.lib.khg_scaled_trans_cost(model.num_tids, _lib.ptr(model.log_probs, C.c_float),
                                              _lib.ptr(model.non_self_loop_log_probs, C.c_float),
                                              _lib.ptr(model.id2state, C.c_int32), _lib.ptr(model.is_self_loop, C.c_uint8),
                                              1.0, 0.1, _lib.ptr(cost, C.c_float)))
    # ONE utterance set, whatever N: every rank builds the whole (feature-less) set from the N = 1 seeds, keeps the
    # utterances shard_utterances deals it, and draws the one global feature stream storing only its own frames
    from kaldi_hmm_gmm_amd.dist import make_comm, shard_utterances, take_utterances
    ut_all = synth.make_utts(model, args.utts, seed=args.seed + 1000, feats=False, transcripts=args.transcripts)
    dev = torch.device("cuda", local)
    if world > 1:
        mine = shard_utterances(np.diff(ut_all.frame_off), world)[rank]
        fo_l, g_l, fr_l = take_utterances(ut_all.frame_off, ut_all.graphs, mine)
        keep = np.zeros(int(ut_all.frame_off[-1]), bool)
        keep[fr_l] = True
        feats = synth.sample_feats_torch(model, ut_all.frame_pdf, args.seed + 2000, dev, keep=keep)
        ut = synth.SynthUtts(fo_l, None, ut_all.ref_ali[fr_l], ut_all.frame_pdf[fr_l], g_l, ut_all.num_phones[mine])
        del keep, fr_l
    else:
        ut = ut_all
        feats = synth.sample_feats_torch(model, ut.frame_pdf, args.seed + 2000, dev)
    n_local = len(ut.frame_off) - 1
    frames_global = int(ut_all.frame_off[-1])
    del ut_all
    torch.cuda.synchronize()

    # One non-null HIP stream shared by torch (events, RCCL) and the khg context.  The shard is cut
    # into batches; the library runs each batch's serial fallback decoder on its own side stream, so
    # that latency-bound tail overlaps the MFMA-bound log-likelihood kernel of the next batch.
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    torch.cuda.set_stream(streams[0])
    ctxs = [Context(local, stream=st.cuda_stream) for st in streams]
    env_k1 = os.environ.get("KHG_K1")
    k1_form = {"fp32": "pdf"}.get(env_k1, env_k1) if env_k1 in ("f16x2s", "f16x2", "pdf", "utt", "fp32") else ("f16x2s" if args.k1 == "auto" else args.k1)
    split_form = k1_form in ("f16x2s", "f16x2")      # fp32 operands split into 16-bit pieces for the 16-bit matrix cores
    for c in ctxs:
        c.set_k1_form(k1_form)
    dm = DeviceModel(ctxs[0], model.gauss_off, gc, model.means_invvars, model.inv_vars)
    tm = DeviceTransitions(ctxs[0], model.id2pdf)
    tm.set_trans_cost(cost)
    nb = max(1, min(args.batches, n_local))
    cuts = [n_local * b // nb for b in range(nb + 1)]
    sets = []
    g = ut.graphs
    for b in range(nb):
        u0, u1 = cuts[b], cuts[b + 1]
        fo = ut.frame_off[u0: u1 + 1]
        so = g["state_off"][u0: u1 + 1]
        ao = g["arc_off"][so[0]: so[-1] + 1]
        sub = {"state_off": so - so[0], "start": g["start"][u0:u1], "arc_off": ao - ao[0],
               "ilabel": g["ilabel"][ao[0]: ao[-1]], "olabel": g["olabel"][ao[0]: ao[-1]],
               "weight": g["weight"][ao[0]: ao[-1]], "nextstate": g["nextstate"][ao[0]: ao[-1]],
               "final": g["final"][so[0]: so[-1]]}
        fsub = feats[int(fo[0]): int(fo[-1])]
        sets.append(UtteranceSet(ctxs[b % len(ctxs)], tm, fo - fo[0], (fsub.data_ptr(), feats), dim=D, graphs=sub))
    accs = DeviceAccs(ctxs[0], dm, tm)
    comm = None
    inproc_fallback = None
    if dist_on and args.allreduce in ("khg", "khg-f32"):
        comm_err = None
        try:
            if os.environ.get("KHG_BENCH_FAIL_COMM") == "2":
                raise RuntimeError("KHG_BENCH_FAIL_COMM=2: pretending ncclCommInitRank failed")
            # (a one-rank group -- KHG_BENCH_FORCE_DIST=1 -- forms a real one-rank RCCL communicator: the same entry points run)
            comm = make_comm(ctxs[0], one_rank=True, timeout_s=float(os.environ.get("KHG_BENCH_COMM_TIMEOUT", "180")))
        except BaseException as ex:
            comm_err = repr(ex)
            print(f"bench.py: rank {rank}/{world} (device {local}): the library's RCCL communicator could not be formed: {comm_err}",
                  file=sys.stderr, flush=True)
        # Under the driver's own torch.distributed.run there is no launch ladder: a rank that cannot form the LIBRARY's communicator must
        # not cost the run.  The ranks agree (over the process group, which is up) whether all of them have it; if not, every rank drops
        # to torch.distributed's all-reduce on a view of the same block -- another code path in the same process, nothing re-executed.
        okt = torch.tensor([0.0 if comm_err else 1.0], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if float(okt[0]) < 0.5:
            inproc_fallback = {"tried": "--allreduce %s (library RCCL communicator)" % args.allreduce, "error": comm_err or "another rank failed",
                               "now": "--allreduce torch"}
            comm = None                      # (a communicator only some ranks hold is left alone: closing it is itself a collective)
            args.allreduce = "torch"
    acc_t = accs.as_torch() if (dist_on and args.allreduce == "torch") else None
    host_block = np.zeros(accs.size, np.float64) if args.allreduce == "host" else None

    T = np.diff(ut.frame_off)
    npdf = np.concatenate([np.diff(s_.pdf_lists()[0]) for s_ in sets])
    frames_local = int(ut.frame_off[-1])
    k1_flops = float((T * npdf).sum()) * (4.0 * D * G + 5.0 * G)  # alignment log-likes
    # cells K1 leaves out with khg_loglikes_reachable: whole 16-frame tiles before a pdf's first readable frame
    skipped_cells = 0.0
    if not args.full_loglikes:
        for s_ in sets:
            poff_, _ = s_.pdf_lists()
            first = s_.pdf_first_frames().astype(np.int64)
            Tu = np.repeat(np.diff(s_.frame_off), np.diff(poff_))
            if k1_form == "f16x2s":                 # whole 32-frame tiles (first needed 32-frame tile, clamped to 255)
                skipped_cells += float(np.minimum(32 * np.minimum(first // 32, 255), Tu).sum())
            elif split_form:                        # whole 32-frame tiles (first needed 16-frame tile, clamped to 127, halved)
                skipped_cells += float(np.minimum(32 * (np.minimum(first // 16, 127) // 2), Tu).sum())
            else:
                skipped_cells += float(np.minimum(16 * (first // 16), Tu).sum())
    band = (not args.full_loglikes) and (not args.no_band) and k1_form == "f16x2s" and args.beam >= 100.0
    if band:                                        # + whole 32-frame tiles past the last useful frame of a pdf
        for s_ in sets:
            poff_, _ = s_.pdf_lists()
            last = s_.pdf_last_frames().astype(np.int64)
            first = s_.pdf_first_frames().astype(np.int64)
            Tu = np.repeat(np.diff(s_.frame_off), np.diff(poff_))
            lt = np.minimum(np.maximum(last, 0) // 32, 254)
            ft = np.minimum(first // 32, 255)
            end = np.minimum(32 * (np.maximum(lt, ft - 1) + 1), Tu)      # cells up to here are computed (or skipped at the front)
            skipped_cells += float((Tu - end).sum())
            # shifted tiles: a band that ends earlier inside its tile than it starts (both known, at least two tiles) is covered by
            # one 32-frame tile fewer
            shifted = (last >= 0) & (ft < 255) & (lt < 254) & (last % 32 < first % 32) & (lt - ft >= 1)
            skipped_cells += 32.0 * float(np.count_nonzero(shifted))
    k1_exec_frac = 1.0 - skipped_cells / max(float((T * npdf).sum()), 1.0)
    args.band_effective = band
    kernel_ms = {}
    ev_a = torch.cuda.Event()
    ev_b = torch.cuda.Event()

    ar_events = []

    dbg = os.environ.get("KHG_BENCH_HOSTDBG") == "1"
    dbg_marks = []

    def step(exchange=True):
        if dbg:
            tt = [time.perf_counter()]
            gm = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            gm[0].record(streams[0])
        accs.zero()                                   # on stream 0
        # One EM iteration changes the parameters once: what the library derives PER PARAMETER VERSION (the fp16 model image, the BAND
        # form's upper bounds, the model's column maxima and scale exponents) is dropped here, so every timed step pays it again
        # as a real iteration does (round 4 left 0.44 ms of it outside the timed region).
        dm.invalidate()
        if dbg: gm[1].record(streams[0])
        ev_a.record(streams[0])
        for st in streams[1:]:
            st.wait_event(ev_a)
        piped = exchange and dist_on and args.allreduce == "khg" and args.c1_parts > 1 and len(streams) == 1
        for s_ in sets:                               # batches alternate between the two streams
            if dbg: tt.append(time.perf_counter())
            s_.loglikes(dm, reachable_only=not args.full_loglikes, band=band)
            if dbg: tt.append(time.perf_counter()); gm[2].record(streams[0])
            s_.align(tm, beam=args.beam, retry_beam=args.retry_beam, acoustic_scale=0.1, download=False)
            if dbg: tt.append(time.perf_counter()); gm[3].record(streams[0])
        for s_ in sets[:-1] if piped else sets:
            s_.acc_stats(dm, tm, accs)
        if dbg:
            tt.append(time.perf_counter()); gm[4].record(streams[0]); dbg_marks.append(gm)
            print("step host ms:", " ".join(f"{(b - a) * 1e3:.2f}" for a, b in zip(tt[:-1], tt[1:])), file=sys.stderr)
        for st in streams[1:]:
            ev_b.record(st); streams[0].wait_event(ev_b)
        if dist_on and exchange:                      # C1, on stream 0 right behind K3: no host synchronisation
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[0])
            if piped:                                 # the last set's K3 with the exchange pipelined behind it by pdf ranges
                sets[-1].acc_stats_reduce(dm, tm, accs, 1.0, comm, args.c1_parts)
            elif acc_t is not None:
                dist.all_reduce(acc_t)                # torch's current stream is stream 0
            elif host_block is not None:
                _lib.check(_lib.lib.khg_accs_download(ctxs[0].h, accs.h, _lib.ptr(host_block, C.c_double)))
                dist.all_reduce(torch.from_numpy(host_block))
                accs.upload(host_block)
            else:
                accs.allreduce(comm, wire_fp32=args.allreduce == "khg-f32")
            e1.record(streams[0])
            ar_events.append((e0, e1))

    # Work done ONCE per utterance set / parameter version, outside the timed region: the column maxima behind K1's scale
    # exponents, the fp16 feature planes, the model's fp16 image.  Its kernels are timed during the first warm-up step.
    prep_ms = {}
    for w in range(args.warmup):
        if w == 0:
            for c in ctxs:
                c.sync(); c.set_timing(True)
        step()
        if w == 0:
            for c in ctxs:
                for name, ms in c.timings():
                    if name in ("k1_absmax", "k1s_pack_x", "k1h_pack_x"):
                        prep_ms[name] = prep_ms.get(name, 0.0) + ms
                c.set_timing(False)
    torch.cuda.synchronize()
    ar_events.clear()
    for c in ctxs:
        c.sync()                                      # also surfaces deferred kernel errors
        c.set_timing(True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    # as timeit does: no cyclic-GC pass inside the timed region (a full collection is several ms of host time; in the first
    # step, when the GPU is idle, it delays the first launch directly -- later steps hide host time behind the running kernels)
    import gc as pygc
    pygc.collect(); pygc.disable()
    t0 = time.perf_counter()
    if dbg:
        marks = [torch.cuda.Event(enable_timing=True)]; marks[0].record(streams[0])
    for _ in range(args.steps):
        step()
        if dbg:
            print(f"issued at {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr)
            marks.append(torch.cuda.Event(enable_timing=True)); marks[-1].record(streams[0])
    torch.cuda.synchronize()
    if dbg:
        print(f"drained at {(time.perf_counter() - t0) * 1e3:.2f} ms; stream-0 marks after each step (ms from the t0 mark): " +
              " ".join(f"{marks[0].elapsed_time(m_):.2f}" for m_ in marks[1:]), file=sys.stderr)
        for gm in dbg_marks[-args.steps:]:
            print("  step on the GPU: from the t0 mark %.2f | zero %.2f  K1 %.2f  K2 %.2f  K3 %.2f" % (marks[0].elapsed_time(gm[0]), gm[0].elapsed_time(gm[1]),
                  gm[1].elapsed_time(gm[2]), gm[2].elapsed_time(gm[3]), gm[3].elapsed_time(gm[4])), file=sys.stderr)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pygc.enable()
    dt_local = dt
    for c in ctxs:
        for name, ms in c.timings():                  # HIP events on the launching stream
            kernel_ms[name] = kernel_ms.get(name, 0.0) + ms
        c.set_timing(False)
        c.sync()
    n_local_launches = args.steps * nb

    frames_total = frames_local
    if dist_on:
        t = torch.tensor([dt, float(frames_local)], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        frames_total = int(t[1])
    # The same shards WITHOUT the exchange, in this very run (round 4 compared with a committed N = 1 line of another box and session):
    # two steps, every rank alone on its shard, slowest rank's time -- what an N-rank run would cost if C1 were free
    alone_ms = None
    if dist_on and world > 1:
        step(exchange=False)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(2):
            step(exchange=False)
        torch.cuda.synchronize()
        tl = torch.tensor([time.perf_counter() - ta], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        alone_ms = float(tl[0]) / 2 * 1e3
        step()                                        # the block holds the sum over all shards again (what `check` and the M-step read)
        torch.cuda.synchronize()
        del ar_events[args.steps:]
        for c in ctxs:
            c.timings()                               # (drained: these steps are not part of kernel_ms)
    # C1 once more, un-pipelined and alone on the stream (nothing to hide behind): what the exchange itself costs at this N
    rccl_info = None
    if dist_on and args.allreduce in ("khg", "khg-f32"):
        c1_alone = None
        if comm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            dist.barrier()
            scratch = DeviceAccs(ctxs[0], dm, tm)     # a block of the same size (zeros): the real one keeps this step's sums
            torch.cuda.synchronize()
            e0.record(streams[0])
            scratch.allreduce(comm)
            e1.record(streams[0])
            torch.cuda.synchronize()
            c1_alone = e0.elapsed_time(e1)
            scratch.close()
        info = comm.info() if comm is not None else {"nranks": 0, "rank": -1, "version": 0}
        v = int(info["version"])
        rccl_info = {"nranks": int(info["nranks"]), "rank": int(info["rank"]), "version_code": v,
                     "version": "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v >= 10000 else str(v),
                     "c1_ms_alone": c1_alone, "c1_bytes": int(accs.size) * 8,
                     "c1_GBps_alone": (int(accs.size) * 8 / (c1_alone * 1e-3) / 1e9) if c1_alone else None,
                     "note": "nranks / rank / version are what RCCL reports for the communicator the exchange ran on (ncclCommCount, "
                             "ncclCommUserRank, ncclGetVersion); c1_ms_alone = one un-pipelined khg_accs_allreduce of the whole fp64 block, "
                             "timed by HIP events after the timed steps"}
    k1_total_ms = kernel_ms.get("k1_loglikes", 0.0)
    k1_avg_ms = k1_total_ms / n_local_launches
    k1_flops_per_launch = k1_flops / nb

    B = types.SimpleNamespace(args=args, ctxs=ctxs, sets=sets, dm=dm, tm=tm, accs=accs, step=step, dist_on=dist_on, dist=dist, dev=dev, backend=backend,
                              nb=nb, k1_form=k1_form, split_form=split_form, frames_total=frames_total, k1_flops_per_launch=k1_flops_per_launch,
                              ar_events=ar_events, n_local=n_local, model=model, gc=gc, D=D)
    fp32_line = bench_lines.fp32_mfma_line(B)
    recipe_line, flat_line = bench_lines.recipe_and_flat_start_lines(B)
    if recipe_line is not None or fp32_line is not None or flat_line is not None:
        step()                                        # the block holds the headline configuration's sums (over all shards) again
        torch.cuda.synchronize()
        del ar_events[args.steps:]
        for c in ctxs:
            c.timings()

    # what every rank did, so that imbalance between the shards is visible on the one line rank 0 prints
    mine_info = {"rank": rank, "utterances": n_local, "frames": frames_local, "seconds": dt_local,
                 "kernel_ms_per_step": {k: v / args.steps for k, v in sorted(kernel_ms.items())}}
    per_rank = [mine_info]
    if dist_on and world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_info)

    if rank == 0:
        tm0 = time.perf_counter()
        res = accs.download()
        tm1 = time.perf_counter()
        # the host M-step of the same iteration, timed separately ("M-step reported separately"):
        # accumulator download -> MleAmDiagGmmUpdate + ComputeGconsts (host C++, fp64) -> new model upload
        from kaldi_hmm_gmm_amd import mle as _mle
        mo = _mle.MleDiagGmmOptions()
        r_up = _mle._flat_update(mo, model.gauss_off, res["occ"], res["mean_acc"], res["var_acc"], 7, 7, model.weights,
                                 model.means_invvars, model.inv_vars)
        tm2 = time.perf_counter()
        dm2 = DeviceModel(ctxs[0], r_up[0], r_up[2], r_up[3], r_up[4])
        ctxs[0].sync()
        tm3 = time.perf_counter()
        dm2.close()
        # the same M-step on the device (K4, khg_model_mle_update): from the accumulators in HBM to a model handle
        # ready for the next pass; checked here against the host result at the bench's full size
        dm.set_weights(model.weights)
        ctxs[0].sync()
        ctxs[0].set_timing(True)
        tm4 = time.perf_counter()
        r_dev = dm.mle_update(accs, mo, 7)
        ctxs[0].sync()
        tm5 = time.perf_counter()
        k4_ms = dict(ctxs[0].timings())
        ctxs[0].set_timing(False)
        d_new = dm.download()
        gc_ulps = np.abs(d_new["gconsts"].astype(np.float64) - r_up[2]) / np.spacing(np.abs(r_up[2]).astype(np.float32))
        m_step = {"host": {"accs_download_ms": (tm1 - tm0) * 1e3, "update_ms": (tm2 - tm1) * 1e3, "model_upload_ms": (tm3 - tm2) * 1e3,
                           "total_ms": (tm3 - tm0) * 1e3},
                  "device": {"total_ms": (tm5 - tm4) * 1e3, "k4_mle_update_ms": k4_ms.get("k4_mle_update"),
                             "k0_pack_tiles_ms": k4_ms.get("k0_pack_tiles"),
                             "hbm_GBps": (int(model.gauss_off[-1]) * (D * 32 + 8)) / max(k4_ms.get("k4_mle_update", 0.0) * 1e-3, 1e-9) / 1e9,
                             "params_bit_equal_to_host": bool(np.array_equal(d_new["weights"], r_up[1]) and
                                                              np.array_equal(d_new["means_invvars"], r_up[3]) and
                                                              np.array_equal(d_new["inv_vars"], r_up[4]) and
                                                              np.array_equal(d_new["gauss_off"], r_up[0])),
                             "gconsts_max_ulps_vs_host": float(gc_ulps.max()),
                             "gconsts_equal_fraction": float((gc_ulps == 0).mean()),
                             "objf_change": r_dev["objf_change"], "host_objf_change": r_up[5]},
                  "gaussians_after": int(r_up[0][-1]),
                  "note": "not part of value (SURVEY 8d: M-step reported separately); host = threaded C++ update between an "
                          "accumulator download and a parameter upload; device = K4 on the accumulators where K3 left them"}
        traffic, traffic_src = pmc_traffic(frames_local / nb, k1_form)
        cells = k1_flops_per_launch / (4.0 * D * G + 5.0 * G)          # (frame, pdf) cells per launch, dense contract
        t_k1 = k1_avg_ms * 1e-3
        if split_form:
            # every fp32 multiply-add of the contraction is NPROD = 3 fp16 multiply-adds (khg_k1_f16x2s.hip.inc, khg_k1_f16x2.hip.inc):
            # the 16-bit FLOPs of the SURVEY 8(d) contract are cells x (NPROD x 4DG + 5G), priced against the dense fp16 MFMA peak
            nprod = 3
            bflops = cells * (nprod * 4.0 * D * G + 5.0 * G)
            kname = {"f16x2s": "k1s_loglikes: v_mfma_f32_32x32x16_f16, f16x2s", "f16x2": "k1h_loglikes: v_mfma_f32_32x32x16_f16, f16x2"}[k1_form]
            roofline = {"bound": "mfma", "kernel": "k1_loglikes (%s)" % kname,
                        "frac_executed": bflops * k1_exec_frac / t_k1 / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                        "achieved": bflops * k1_exec_frac / t_k1 / 1e12,
                        "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": bflops * k1_exec_frac / t_k1 / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                        "frac_dense_contract": bflops / t_k1 / 1e12 / PEAK_BF16_MFMA_TFLOPS, "achieved_dense_contract": bflops / t_k1 / 1e12,
                        "fp32_equivalent": {"achieved": k1_flops_per_launch / t_k1 / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                            "frac": k1_flops_per_launch / t_k1 / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                            "note": "the SURVEY 8(d) fp32 contract FLOPs (4DG + 5G per cell) over the same time, against the fp32-MFMA "
                                                    "peak the fp32 kernels are bound by: > 1 means past that roofline"},
                        "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                        "executed_cell_fraction": k1_exec_frac,
                        "k1_cells": "band: per pdf from its first readable to its last useful 32-frame tile, the tiles starting at the band's first frame where that saves one (khg_loglikes_band)" if band else
                                    ("all" if args.full_loglikes else "from each pdf's first readable tile (khg_loglikes_reachable)"),
                        "note": "frac = frac_executed = achieved / peak: the 16-bit FLOPs of the cells K1 EXECUTES (%d partial products per fp32 "
                                "product; what SQ_VALU_MFMA_BUSY shows) / kernel time / the 2.5 PFLOP/s dense fp16 = bf16 peak -- the utilisation.  "
                                "frac_dense_contract credits the whole T x P_u contract of SURVEY 8(d), i.e. also the cells K1 skips because no "
                                "decoder token can read them (executed_cell_fraction of the contract is executed, in whole 32-frame "
                                "tiles): work avoided, not utilisation.  K1 is bound by POWER: under a bare "
                                "dependent 16-bit MFMA loop the chip holds 1.5-1.8 GHz (18.7-21.5 ns per v_mfma_f32_32x32x16 and SIMD on the boxes "
                                "of this pool, tools/k1lab.hip: 0.60-0.72 of the 2.5 PFLOP/s spec peak), and every byte moved and VALU "
                                "instruction issued beside the MFMAs lowers the clock further" % nprod,
                        "kernel_ms": k1_avg_ms, "flops_per_launch": bflops, "launches_per_step": nb}
        else:
            ach = k1_flops_per_launch / t_k1 / 1e12
            roofline = {"bound": "mfma", "kernel": "k1_loglikes (fp32 MFMA, %s)" % k1_form, "frac_executed": ach * k1_exec_frac / PEAK_F32_MFMA_TFLOPS,
                        "achieved": ach * k1_exec_frac, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach * k1_exec_frac / PEAK_F32_MFMA_TFLOPS, "frac_dense_contract": ach / PEAK_F32_MFMA_TFLOPS, "achieved_dense_contract": ach,
                        "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                        "executed_cell_fraction": k1_exec_frac,
                        "note": "frac = frac_executed: the FLOPs of the (frame, pdf) cells K1 executes (those a decoder token can read, in whole "
                                "16-frame tiles: executed_cell_fraction of the dense T x P_u contract of SURVEY 8(d)) -- the MFMA utilisation the "
                                "SQ_VALU_MFMA_BUSY_CYCLES counter shows; frac_dense_contract credits the whole contract",
                        "kernel_ms": k1_avg_ms, "flops_per_launch": k1_flops_per_launch, "launches_per_step": nb}
        # "K1 vs MFMA peak, K2 / K3 vs HBM peak, C1 vs 7 x 153 GB/s" -- the bandwidth-bound kernels by their ALGORITHMIC bytes
        # per launch (what has to cross HBM once) over the HIP-event time of the launch, against 8 TB/s
        PEAK_HBM_GBPS, PEAK_XGMI_GBPS = 8000.0, 7 * 153.0
        S_u = np.diff(ut.graphs["state_off"]).astype(np.float64)
        tpad = ((T + 31) // 32 * 32).astype(np.float64)
        k2_bytes = float((npdf * tpad).sum()) * 4 + float((T * S_u).sum()) * 0.5 * 2 + float(T.sum()) * 4
        sumG_ = float(model.gauss_off[-1])
        k3_bytes = float(T.sum()) * (4.0 * D + 8.0) + sumG_ * (2 * D + 1) * (4.0 + 8.0)
        kms = {k: v / args.steps for k, v in kernel_ms.items()}

        def hbm(bytes_, ms):
            return {"bytes": bytes_, "ms": ms, "GBps": bytes_ / (ms * 1e-3) / 1e9 if ms else None,
                    "frac_hbm": bytes_ / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS if ms else None}
        roofline["kernels"] = {
            "k2": dict(hbm(k2_bytes, kms.get("k2_viterbi_dp", 0.0)), kernel="k2_viterbi_dp",
                       bytes_note="scores read once (4 B x listed pdfs x padded frames) + back-pointers written and read (a nibble per state and "
                                  "frame, twice) + the alignment (4 B per frame)"),
            "k3": dict(hbm(k3_bytes, kms.get("k3_accumulate", 0.0)), kernel="k3_accumulate",
                       bytes_note="features gathered once (4 D B per frame) + alignment and frame ids (8 B per frame) + the model rows read and the "
                                  "fp64 accumulator rows flushed once (12 B x (2 D + 1) per Gaussian)"),
            "k3_bucket": dict(hbm(float(T.sum()) * (4.0 + 3 * 12.0), kms.get("k3_bucket", 0.0)), kernel="k3_sort_keys + radix sort + k3_bounds",
                              bytes_note="alignment read (4 B) + key / value pairs written, sorted (one read + one write per pair) and read back"),
        }
        if dist_on and world > 1:
            c1_ms = kms.get("c1_allreduce", 0.0)
            blk = float(accs.size) * 8
            bus = 2.0 * (world - 1) / world * blk
            roofline["kernels"]["c1"] = {"bytes": blk, "ms_pipelined_pieces": c1_ms, "ms_alone": rccl_info["c1_ms_alone"] if rccl_info else None,
                                         "busbw_GBps": bus / (rccl_info["c1_ms_alone"] * 1e-3) / 1e9 if rccl_info and rccl_info.get("c1_ms_alone") else None,
                                         "frac_xgmi": bus / (rccl_info["c1_ms_alone"] * 1e-3) / 1e9 / PEAK_XGMI_GBPS if rccl_info and rccl_info.get("c1_ms_alone") else None,
                                         "note": "ring all-reduce: 2 (N - 1) / N x block bytes leave every GPU; against 7 links x 153 GB/s"}
        out = {
            "metric": "frames/sec (whole node) per EM iter (align+acc-stats), 5k-pdf x 64-Gauss",
            "value": frames_total * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if not split_form else "f32 via " + k1_form,
            "dtype_note": None if not split_form else (
                "log-likelihood contraction: fp32 operands rescaled per k by an exact power of two and split into two fp16 pieces (11 + 11 "
                "significant bits: |v - (v1 + v2)| <= max(2^-23 |v|, 2^-25) in scaled units), the three partial products w1 x1, w1 x2, "
                "w2 x1 on the fp16 matrix cores into ONE fp32 accumulator; the dropped w2 x2 is <= 2^-22 of the term, worst case 2^-21 per "
                "term plus an absolute floor the library keeps under 2e-6 (else it runs the two-accumulator f16x2 form, else fp32 MFMA); "
                "measured error vs fp64: max 2.1e-7 B (tools/k1lab.hip; the fp32 fmaf chain of the fp32-MFMA kernels: 7.4e-7 B), every "
                "fp64-bound tolerance test (1e-5 + 1e-6 B) unchanged; everything else fp32 / fp64 as the reference; the fp32-MFMA K1 is "
                "timed beside it in fp32_mfma_line (--k1 pdf runs it as the whole bench)" if k1_form == "f16x2s" else
                "log-likelihood contraction: fp32 operands rescaled per k by an exact power of two and written as v1 + v2 2^-11 with two fp16 "
                "pieces (|v - (v1 + v2 2^-11)| <= 2^-23 |v|: 11 + 11 significant bits), the three partial products w1 x1, w1 x2, w2 x1 on the "
                "fp16 matrix cores with fp32 accumulators; the dropped w2 x2 is <= 2^-22 of the term -- worst case 2^-21 per term, measured "
                "error vs fp64 max 3.0e-7 B (the fp32 fmaf chain: 7.4e-7 B, profiles/r2_probe_f16x2.txt), every fp64-bound tolerance test "
                "unchanged; everything else fp32 / fp64 as the reference; the fp32-MFMA K1 is timed beside it in fp32_mfma_line "
                "(--k1 pdf runs it as the whole bench)" if k1_form == "f16x2" else
                "fp32 MFMA (v_mfma_f32_16x16x4_f32): the reference's per-Gaussian fmaf chain bit for bit"),
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} pdfs x {G} Gauss, dim {D}, {args.utts} utterances "
                                   f"({frames_total} frames) sharded over {world} GPU(s), beam {args.beam:g}, "
                                   f"acoustic_scale 0.1, mean pdfs/utt {npdf.mean():.1f}" + (", Zipf-lexicon transcripts" if args.transcripts == "zipf" else ""),
                       "frames_per_step": frames_total, "utterances": args.utts},
            "roofline": roofline,
            "fp32_mfma_line": fp32_line,
            "kernel_ms_per_step": {k: v / args.steps for k, v in sorted(kernel_ms.items())},
            "prep_ms": {"kernels": prep_ms, "total": sum(prep_ms.values()),
                        "note": "once per UTTERANCE SET (the features' column maxima -- k1_absmax here also covers the model's columns of the "
                                "first step --, the fp16 feature planes): outside the timed region, measured during the first warm-up step.  What "
                                "is derived per PARAMETER VERSION (k0_model_stats: column maxima, feature envelope, band upper bounds in one pass; k0s_pack_tiles) is INSIDE "
                                "every timed step since round 5 (khg_model_invalidate at the top of the step) and listed in kernel_ms_per_step"},
            "allreduce_ms_per_step": (sum(a_.elapsed_time(b_) for a_, b_ in ar_events) / max(len(ar_events), 1)) if ar_events else None,
            "allreduce_bytes": int(accs.size) * 8 if dist_on else None,
            "m_step": m_step,
            "check": {"acc_total_frames": res["total_frames"], "frames_in_set": frames_global, "avg_loglike_per_frame":
                      res["total_log_like"] / max(res["total_frames"], 1.0)},
            "allreduce": args.allreduce if dist_on else None,
            "allreduce_fallback_in_process": inproc_fallback,
            "rccl": rccl_info,
            "scaling_efficiency_vs_n1_shard": None if alone_ms is None else {
                "value": alone_ms / (dt / args.steps * 1e3), "shards_without_exchange_ms_per_step": alone_ms, "ms_per_step": dt / args.steps * 1e3,
                "note": "measured in THIS run: two extra steps with every rank alone on its shard (no C1), slowest rank, over the timed "
                        "ms_per_step -- 1.0 = the exchange is fully hidden; what is left of the N-fold speed-up besides this is the small-shard "
                        "effect (fewer, shorter launches per kernel), visible as N x shards_without_exchange_ms_per_step against the N = 1 line"},
            "c1_pipelined_parts": args.c1_parts if (dist_on and args.allreduce == "khg" and args.c1_parts > 1 and len(streams) == 1) else None,
            "per_rank": per_rank if world > 1 else None,
        }
        if out["c1_pipelined_parts"]:
            out["allreduce_note"] = ("allreduce_ms_per_step spans the LAST set's K3 with the exchange pipelined behind it in %d pdf ranges "
                                     "(khg_acc_stats_reduce); the RCCL pieces alone are kernel_ms_per_step['c1_allreduce']" % args.c1_parts)
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline belongs to the N = 1 line only: at N > 1 the other ranks would sit
                                                          # at the closing barrier for its 15 s (and the oracle check that rides on it)
            ncpu = min(n_local, 40000)     # enough work for a few seconds of every host core
            fh = feats[: int(ut.frame_off[ncpu])].cpu().numpy()
            ans, out["cpu_baseline"] = cpu_baseline(model, gc, ut, cost, {"n": ncpu, "feats": fh}, args.cpu_baseline_seconds,
                                                    beam=args.beam, retry_beam=args.retry_beam)
            # the oracle's alignments and accumulators of that sample against the product's, at the benchmark's own shape
            out["check"].update(check_vs_oracle(ans, ut, feats, D, sets, ctxs[0], model, gc, tm, args))
        else:
            out["cpu_baseline"] = None
        out["recipe_beam_line"] = recipe_line
        out["flat_start_line"] = flat_line
        if args.per_call_utts > 0 and world == 1:
            out["per_call_line"] = per_call_line(args, model, ut, feats, D, ctxs[0], out.get("cpu_baseline"))
        record = json.dumps(out)
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (record + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
