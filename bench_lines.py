#!/usr/bin/env python3
"""bench_lines.py -- everything bench.py prints BESIDE its timed step: the self-launch ladder of `python bench.py --gpus N`, the CPU baseline
(the oracle on the host's cores), the oracle check at the benchmark's scale, the per-utterance call pattern, the fp32-MFMA line, the
recipe-beam line and the flat-start line, the HBM-traffic figures of the committed counter pass.  bench.py itself holds the argument
parser, the workload, the timed step and the assembly of the ONE JSON line (round 6: split out of a 1100-line bench.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: N fresh child processes, one per GPU, under torch.distributed.run.  This process
    has imported neither torch nor the library: nothing here has initialised a GPU, and nothing is exec'ed -- the children are
    ordinary subprocesses.  The first N > 1 run must not be wasted on one broken piece of the exchange, so the launch walks a LADDER,
    every rung in FRESH children (never a retry inside a process that has touched a GPU):
      0. --dist-selftest with the requested exchange: process group + communicator + a 1 MB all-reduce, seconds, no data built;
      1. the bench as asked (--allreduce khg, C1 pipelined behind K3 by the library's own RCCL calls);
      2. --allreduce torch (torch.distributed's all-reduce on a view of the block);
      3. --allreduce khg --c1-parts 1 (one un-pipelined all-reduce by the library).
    What failed on the way is recorded in the line that finally prints ("allreduce_fallback")."""
    import socket
    import subprocess

    def run(extra, timeout):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        base = [a for a in sys.argv[1:]]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + base + extra
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the launcher and its ranks form their own process group: a rung that times out is ended as a GROUP (the elastic agent, if
        # killed alone, cannot reap ranks hung in RCCL -- they would keep the GPUs and the pipes while the next rung starts)
        import signal
        pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, start_new_session=True)
        try:
            o, e = pr.communicate(timeout=timeout)
            rc, out, err = pr.returncode, o.decode("utf-8", "replace"), e.decode("utf-8", "replace")
        except subprocess.TimeoutExpired:
            for sig, grace in ((signal.SIGTERM, 20), (signal.SIGKILL, 20)):
                try:
                    os.killpg(pr.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    pr.wait(timeout=grace)
                    break
                except subprocess.TimeoutExpired:
                    continue
            try:
                os.killpg(pr.pid, signal.SIGKILL)      # ranks that outlived the agent
            except ProcessLookupError:
                pass
            try:
                o, e = pr.communicate(timeout=20)
            except Exception:
                o, e = b"", b""
            rc, out, err = 124, o.decode("utf-8", "replace"), e.decode("utf-8", "replace") + f"\n[self_launch] timed out after {timeout} s"
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        return rc, (lines[-1] if lines else None), err

    asked = args.allreduce
    rungs = [("as asked (--allreduce %s, --c1-parts %d)" % (asked, args.c1_parts), [])]
    if asked == "khg":
        rungs += [("--allreduce torch", ["--allreduce", "torch"]), ("--allreduce khg --c1-parts 1", ["--allreduce", "khg", "--c1-parts", "1"])]
    tried = []
    t_self = float(os.environ.get("KHG_BENCH_SELFTEST_TIMEOUT", "900"))      # (the first `import torch` on a fresh box alone can take minutes)
    t_run = float(os.environ.get("KHG_BENCH_RUN_TIMEOUT", "3000"))
    for k, (name, extra) in enumerate(rungs):
        if not args.dist_selftest:
            rc, line, err = run(extra + ["--dist-selftest"], t_self)
            if rc or not line:
                tried.append({"tried": name + " [selftest]", "rc": rc, "stderr_tail": [ln for ln in err.splitlines() if ln.startswith("bench.py:")][-4:] + err.splitlines()[-8:]})
                print(f"bench.py: dist selftest failed on rung {k} ({name}), rc {rc}; trying the next rung", file=sys.stderr)
                continue
        rc, line, err = run(extra, t_run)
        if rc == 0 and line:
            if tried:
                try:
                    d = json.loads(line)
                    d["allreduce_fallback"] = tried
                    line = json.dumps(d)
                except Exception:
                    pass
            print(line, flush=True)
            sys.stderr.write(err[-2000:])
            sys.exit(0)
        tried.append({"tried": name, "rc": rc, "stderr_tail": [ln for ln in err.splitlines() if ln.startswith("bench.py:")][-4:] + err.splitlines()[-8:]})
        print(f"bench.py: the {args.gpus}-rank launch on rung {k} ({name}) ended with return code {rc}; last lines of its stderr:", file=sys.stderr)
        for ln in err.splitlines()[-60:]:
            print("  | " + ln, file=sys.stderr)
        if args.dist_selftest:
            break
    print(json.dumps({"metric": "frames/sec (whole node) per EM iter (align+acc-stats), 5k-pdf x 64-Gauss", "value": None, "n_gpus": args.gpus,
                      "error": "every rung of the launch ladder failed", "allreduce_fallback": tried}), flush=True)
    sys.exit(3)


def csrc_sha():
    """Identity of the kernels a PMC profile belongs to: sha256 over the library's sources."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kaldi_hmm_gmm_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".inc")):            # the device code and its launch code; the host classes do not touch the kernels
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fn.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(frames_per_launch, k1_form="f16x2s"):
    """HBM-side bytes per K1 launch from the committed rocprofv3 PMC passes (profiles/r*_pmc_summary.json: separate
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction) -- bench.py cannot run the profiler on
    itself.  Only reported when the profile was taken at this launch size AND on these kernel sources (csrc_sha);
    otherwise null with the reason."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            with open(path) as fh:
                pm = json.load(fh)
            want = {"f16x2s": "k1s_loglikes", "f16x2": "k1h_loglikes", "pdf": "k1p_loglikes", "utt": "k1_loglikes"}[k1_form]
            k1 = next(v for k, v in pm["kernels"].items() if k.startswith(want))
            rel = os.path.relpath(path, ROOT)
            if abs(pm["frames_per_launch"] / frames_per_launch - 1.0) > 0.02:
                return None, f"{rel}: collected at another launch size ({pm['frames_per_launch']} frames)"
            if pm.get("csrc_sha") != csrc_sha():
                return None, f"{rel}: collected on other kernel sources (csrc_sha {pm.get('csrc_sha')}, now {csrc_sha()})"
            return k1["traffic_bytes"], f"{rel} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same launch size, csrc_sha {pm['csrc_sha']})"
        except Exception:
            continue
    return None, "no PMC summary under profiles/"


def cpu_baseline(model, gc, ut, cost, feats_host, budget_s, beam=200.0, retry_beam=0.0):
    """The oracle (CPU restatement, kind="port") timed on a bounded sample of the SAME workload:
    AlignUtteranceWrapper + acc-stats per utterance, one thread -- the reference's execution model."""
    from oracle import oracle as orc

    om = orc.OModel(model.gauss_off, gc, model.means_invvars, model.inv_vars)
    g = dict(ut.graphs)
    g["weight"] = np.where(g["ilabel"] >= 1, g["weight"] + cost[g["ilabel"]], g["weight"]).astype(np.float32)
    nmax = feats_host["n"]

    kept = {"ali": [], "status": [], "like": []}          # the oracle's answers, for the parity check at this scale (check_vs_oracle)

    def one(u, oa):
        og = orc.OGraph.from_set(g, u)
        f = feats_host["feats"][ut.frame_off[u]: ut.frame_off[u + 1]]
        r = orc.align_utterance(og, om, model.id2pdf, f, acoustic_scale=0.1, beam=beam, retry_beam=retry_beam)
        ok = (r["status"] & 1) == 0
        if ok:
            orc.acc_stats_ali(om, model.id2pdf, f, r["ali"], oa)
        kept["ali"].append(np.asarray(r["ali"], np.int32) if ok else np.zeros(f.shape[0], np.int32))
        kept["status"].append(int(r["status"])); kept["like"].append(np.float32(r["like"]) if ok else np.float32(0))
        return f.shape[0]

    # (A) one thread: the reference's execution model (its scripts loop over utterances in Python), built with the
    # reference's default Release flags (-O3, no -march: BASELINE.md section 3 (i))
    orc.use("o3")
    oa = orc.OAccs(int(model.gauss_off[-1]), model.dim, model.num_tids)
    frames1 = n1 = 0
    t0 = time.perf_counter()
    while n1 < nmax and (time.perf_counter() - t0 < budget_s / 2 or n1 < 4):
        frames1 += one(n1, oa)
        n1 += 1
    dt1 = time.perf_counter() - t0
    # (B) utterance-parallel over ALL host cores inside the C oracle (orc_em_pass_mt: POSIX threads, a private accumulator
    # set per thread, the same two calls per utterance), built -O3 -march=native
    orc.use("native")
    # bounded by memory: every thread owns an accumulator set (sumG * (2 D + 1) doubles)
    acc_bytes = int(model.gauss_off[-1]) * (2 * model.dim + 1) * 8
    try:
        import psutil
        mem_cap = int(0.25 * psutil.virtual_memory().available // max(acc_bytes, 1))
    except Exception:
        mem_cap = 32
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = ncpu                       # a container may see every core of the host but be allowed only a few of them
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:               # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()
            if q != "max":
                quota = max(1, -(-int(q) // int(per)))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:   # cgroup v1
                q, per = int(f1.read()), int(f2.read())
                if q > 0:
                    quota = max(1, -(-q // per))
        except Exception:
            pass
    nthr = max(1, min(ncpu, quota, mem_cap, 512))
    if os.environ.get("KHG_BENCH_CPU_THREADS"):
        nthr = max(1, int(os.environ["KHG_BENCH_CPU_THREADS"]))
    keep_b = {}
    fr, nn, failed, dtb = orc.em_pass_mt(om, model.id2pdf, g, ut.frame_off, feats_host["feats"], first_utt=n1, n_utt=max(nmax - n1, 0),
                                         num_threads=nthr, budget_seconds=budget_s / 2, acoustic_scale=0.1, beam=beam, retry_beam=retry_beam,
                                         keep=keep_b)
    done = [fr, nn]
    orc.use(None)
    # what both legs computed, as one answer over utterances [0, n1 + nn): alignments, status, like, accumulators (A + B)
    nb_fr = int(ut.frame_off[n1 + nn] - ut.frame_off[n1])
    ob = keep_b["accs"]
    oracle_answer = {"n_utt": n1 + nn,
                     "ali": np.concatenate(kept["ali"] + [keep_b["ali"][:nb_fr]]) if (n1 + nn) else np.zeros(0, np.int32),
                     "status": np.concatenate([np.asarray(kept["status"], np.int32), keep_b["status"][:nn]]),
                     "like": np.concatenate([np.asarray(kept["like"], np.float32), keep_b["like"][:nn]]),
                     "occ": oa.occ + ob.occ, "mean_acc": oa.mean_acc + ob.mean_acc, "var_acc": oa.var_acc + ob.var_acc,
                     "trans_acc": oa.trans_acc + ob.trans_acc, "total_frames": oa.total_frames + ob.total_frames,
                     "total_log_like": oa.total_log_like + ob.total_log_like}
    par = done[0] / dtb if done[1] else 0.0
    one_thr = frames1 / dt1
    best, cores = (par, nthr) if par > one_thr else (one_thr, 1)
    return oracle_answer, {"value": best, "unit": "frames/s", "cores": cores, "kind": "port", "one_thread_value": one_thr,
            "sample": f"oracle/khg_oracle.c: FasterDecoder + GMM decodable + acc-stats per utterance of rank 0's shard; "
                      f"1 thread (gcc -O3, the reference's default flags): first {n1} utterances ({frames1} frames) in {dt1:.1f}s; "
                      f"{nthr} POSIX threads inside the C oracle ({ncpu} logical CPUs visible, CPU quota {quota}; gcc -O3 -march=native, "
                      f"utterance-parallel, private accumulators): "
                      f"next {done[1]} utterances ({done[0]} frames) in {dtb:.1f}s"}


def check_vs_oracle(ans, ut, feats, D, sets, ctx, model, gc, tm, args):
    """Parity at the benchmark's scale, after the timed region: the product's alignment of the utterances the CPU baseline
    just aligned (K1 + K2 once more with the results downloaded) against the oracle's, utterance by utterance, and K3 over exactly
    those utterances against the oracle's accumulators (tolerances of tests/test_gpu_parity.py: rtol 2e-5)."""
    from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, UtteranceSet

    n = int(ans["n_utt"])
    if n == 0:
        return {"oracle_utts": 0}
    # the parameters the timed steps (and the oracle) used: the bench's own model handle went through the M-step since
    dm = DeviceModel(ctx, model.gauss_off, gc, model.means_invvars, model.inv_vars)
    nfr = int(ut.frame_off[n])
    got_ali, got_status, got_like = [], [], []
    for s_ in sets:                                   # the sets cover the shard in utterance order
        if sum(len(x) for x in got_status) >= n:
            break
        s_.loglikes(dm, reachable_only=not args.full_loglikes, band=args.band_effective)
        r = s_.align(tm, beam=args.beam, retry_beam=args.retry_beam, acoustic_scale=0.1, download=True)
        got_ali.append(np.asarray(r["ali"])); got_status.append(np.asarray(r["status"])); got_like.append(np.asarray(r["like"]))
    g_ali = np.concatenate(got_ali)[:nfr]; g_status = np.concatenate(got_status)[:n]; g_like = np.concatenate(got_like)[:n]
    o_ali, o_status, o_like = ans["ali"][:nfr], ans["status"][:n], ans["like"][:n]
    neq = g_ali != o_ali
    bad_utts = np.add.reduceat(neq.astype(np.int64), ut.frame_off[:n].astype(np.int64)) > 0 if nfr else np.zeros(n, bool)
    bad_utts &= np.diff(ut.frame_off[: n + 1]) > 0
    status_mismatch = int((((g_status & 1) != 0) != ((o_status & 1) != 0)).sum() + (((g_status & 2) != 0) != ((o_status & 2) != 0)).sum())
    okm = ((g_status & 1) == 0) & ((o_status & 1) == 0)
    like_rel = float(np.max(np.abs(g_like[okm] - o_like[okm]) / np.maximum(np.abs(o_like[okm]), 1.0))) if okm.any() else 0.0
    # K3 on exactly those utterances, from the ORACLE's alignment (so that a decoder difference cannot hide in the statistics)
    sub = UtteranceSet(ctx, None, ut.frame_off[: n + 1].astype(np.int64), (feats.data_ptr(), feats), dim=D)
    sub.upload_ali(np.ascontiguousarray(o_ali, np.int32))
    acc2 = DeviceAccs(ctx, dm, tm)
    sub.acc_stats(dm, tm, acc2)
    got = acc2.download()
    sub.close(); acc2.close(); dm.close()

    def rel(a, b, atol):
        return float(np.max(np.abs(a - b) / (np.abs(b) + atol)))
    mmax, vmax = float(np.abs(ans["mean_acc"]).max()), float(np.abs(ans["var_acc"]).max())
    e_occ, e_mean, e_var = rel(got["occ"], ans["occ"], 1e-6 / 2e-5), rel(got["mean_acc"], ans["mean_acc"], 2e-6 * mmax / 2e-5), \
        rel(got["var_acc"], ans["var_acc"], 2e-6 * vmax / 2e-5)
    return {"oracle_utts": n, "oracle_frames": nfr, "ali_mismatch_utts": int(bad_utts.sum()), "ali_mismatch_frames": int(neq.sum()),
            "status_mismatch": status_mismatch, "oracle_failed_utts": int(((o_status & 1) != 0).sum()),
            "like_max_rel_err": like_rel,
            "max_rel_err_occ": e_occ, "max_rel_err_mean_acc": e_mean, "max_rel_err_var_acc": e_var,
            "trans_acc_equal": bool(np.array_equal(got["trans_acc"], ans["trans_acc"])),
            "stats_within_2e-5": bool(max(e_occ, e_mean, e_var) <= 2e-5),
            "total_frames_equal": bool(got["total_frames"] == ans["total_frames"]),
            "avg_loglike_per_frame_oracle": ans["total_log_like"] / max(ans["total_frames"], 1.0),
            "avg_loglike_per_frame_k3": got["total_log_like"] / max(got["total_frames"], 1.0),
            "note": "after the timed region: the utterances bench.py's cpu_baseline aligned with oracle/khg_oracle.c (FasterDecoder + GMM "
                    "decodable + acc-stats) against K1 + K2 (alignment, status, like per utterance) and K3 (statistics of those utterances "
                    "from the oracle's alignment); max_rel_err_* = max |got - want| / (|want| + atol) with the atol of tests/test_gpu_parity.py "
                    "folded in (<= 2e-5 passes)"}


def per_call_line(args, model, ut, feats, D, ctx, cpu_base):
    """The reference's OWN call pattern at the benchmark's model size (egs/yesno/train.py:170-202): one gmm_align_compiled(...) and one
    gmm_acc_stats_ali(...) per utterance, the reference's keyword arguments, host numpy features and a StdVectorFst copy per call --
    after the timed region.  Alignments must equal the batched path's, statistics agree within 2e-5."""
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd import _gpu, synth

    n = int(min(args.per_call_utts, len(ut.frame_off) - 1))
    nw = min(8, n // 4)                                     # warm-up calls: the model's upload, the fp16 image settling on the exponents
    _gpu.set_default_context(ctx)
    t0 = time.perf_counter()
    am, tmh = synth.host_objects(model)
    t_host = time.perf_counter() - t0
    fh = feats[: int(ut.frame_off[n])].cpu().numpy()
    fl = [np.ascontiguousarray(fh[int(ut.frame_off[u]): int(ut.frame_off[u + 1])]) for u in range(n)]
    fsts = [synth.utt_fst(ut.graphs, u) for u in range(n)]
    cfg = khg.AlignConfig(beam=args.beam, retry_beam=args.retry_beam, careful=False)

    def one(u, gmm_accs, tacc):
        ans = khg.gmm_align_compiled(am_gmm=am, transition_model=tmh, utt=str(u), fst=fsts[u].copy(), feats=fl[u], align_config=cfg,
                                     acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
        t1 = time.perf_counter()
        ll, tacc = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=gmm_accs, transition_model=tmh, feats=fl[u], ali=ans["alignment"], transition_accs=tacc)
        return ans, ll, tacc, t1

    warm = khg.AccumAmDiagGmm(); warm.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    t0 = time.perf_counter()
    one(0, warm, None)
    t_first = time.perf_counter() - t0
    for u in range(1, nw):
        one(u, warm, None)
    del warm
    gmm_accs = khg.AccumAmDiagGmm(); gmm_accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    tacc, alis, t_align, t_acc, done, tot_ll = None, [], 0.0, 0.0, 0, 0.0
    import gc as pygc
    pygc.collect(); pygc.disable()
    t_begin = time.perf_counter()
    for u in range(n):
        ta = time.perf_counter()
        ans, ll, tacc, t1 = one(u, gmm_accs, tacc)
        tb = time.perf_counter()
        t_align += t1 - ta; t_acc += tb - t1
        alis.append(np.asarray(ans["alignment"], np.int32)); done += ans["num_done"]; tot_ll += ll
    dt = time.perf_counter() - t_begin
    pygc.enable()
    frames = int(ut.frame_off[n])
    t0 = time.perf_counter()
    tot_count = gmm_accs.tot_count                      # the first host-side read: the device statistics come down once
    t_flush = time.perf_counter() - t0
    # the batched path on the same utterances, same parameters
    go, gc_, _, miv, iv = am.flat()
    dmb = khg.DeviceModel(ctx, go, gc_, miv, iv)
    tmb = khg.DeviceTransitions(ctx, np.asarray(tmh.transition_id_to_pdf_array(), np.int32))
    tmb.set_trans_cost(np.asarray(tmh.scaled_trans_cost(1.0, 0.1), np.float32))
    g = ut.graphs
    so = g["state_off"][: n + 1]
    ao = g["arc_off"][: so[-1] + 1]
    sub = {"state_off": so, "start": g["start"][:n], "arc_off": ao, "ilabel": g["ilabel"][: ao[-1]], "olabel": g["olabel"][: ao[-1]],
           "weight": g["weight"][: ao[-1]], "nextstate": g["nextstate"][: ao[-1]], "final": g["final"][: so[-1]]}
    us = khg.UtteranceSet(ctx, tmb, ut.frame_off[: n + 1].astype(np.int64), fh, graphs=sub)
    us.loglikes(dmb, reachable_only=True)
    res = us.align(tmb, beam=args.beam, retry_beam=args.retry_beam, acoustic_scale=0.1)
    accb = khg.DeviceAccs(ctx, dmb, tmb)
    us.acc_stats(dmb, tmb, accb)
    st = accb.download()
    accb.close(); us.close(); tmb.close(); dmb.close()
    ali_equal = bool(np.array_equal(np.concatenate(alis), np.asarray(res["ali"])))
    P = model.num_pdfs
    occ = np.concatenate([np.asarray(a.occupancy) for a in gmm_accs._accs])
    mean = np.concatenate([np.asarray(a.mean_accumulator).ravel() for a in gmm_accs._accs])
    var = np.concatenate([np.asarray(a.variance_accumulator).ravel() for a in gmm_accs._accs])

    def rel(a, b, atol):
        return float(np.max(np.abs(a - b) / (np.abs(b) + atol)))
    mm, vm = float(np.abs(st["mean_acc"]).max()), float(np.abs(st["var_acc"]).max())
    errs = (rel(occ, st["occ"], 1e-6 / 2e-5), rel(mean, st["mean_acc"].ravel(), 2e-6 * mm / 2e-5), rel(var, st["var_acc"].ravel(), 2e-6 * vm / 2e-5))
    one_thr = (cpu_base or {}).get("one_thread_value")
    return {"value": frames / dt, "unit": "frames/s", "utterances": n, "frames": frames, "warmup_calls": nw,
            "ms_per_utt": dt / n * 1e3, "ms_per_utt_align": t_align / n * 1e3, "ms_per_utt_acc_stats": t_acc / n * 1e3,
            "first_call_ms": t_first * 1e3, "host_objects_s": t_host, "accumulator_flush_ms": t_flush * 1e3,
            "num_done": int(done), "ali_identical_to_batched": ali_equal,
            "max_rel_err_occ": errs[0], "max_rel_err_mean_acc": errs[1], "max_rel_err_var_acc": errs[2], "stats_within_2e-5": bool(max(errs) <= 2e-5),
            "tot_count": float(tot_count), "sum_log_like": tot_ll, "batched_total_log_like": st["total_log_like"],
            "vs_cpu_one_thread": (frames / dt) / one_thr if one_thr else None,
            "note": "after the timed region: gmm_align_compiled(am_gmm=, transition_model=, utt=, fst=<copy>, feats=<host numpy>, align_config=, "
                    "acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1) + gmm_acc_stats_ali(am_gmm=, gmm_accs=, transition_model=, "
                    "feats=, ali=, transition_accs=) once per utterance (egs/yesno/train.py:170-202); the model and the transition table are "
                    "cached on the device by the host objects' mutation version (first_call_ms holds the one upload), the statistics stay on "
                    "the device until something reads them (accumulator_flush_ms); vs_cpu_one_thread = value / cpu_baseline.one_thread_value"}


def fp32_mfma_line(B):
    """The same step with K1 switched to the pdf-major fp32-MFMA kernel (the reference's fmaf chain bit for bit): 1 warm-up + 2 timed steps.
    B: bench.py's namespace (args, ctxs, sets, step, dist / torch handles, frames_total, ...)."""
    import torch
    args, ctxs, step, dist_on, dist, dev, backend, nb, k1_form = B.args, B.ctxs, B.step, B.dist_on, B.dist, B.dev, B.backend, B.nb, B.k1_form
    frames_total, k1_flops_per_launch, ar_events, split_form = B.frames_total, B.k1_flops_per_launch, B.ar_events, B.split_form
    # the fp32-MFMA K1 beside a split-form run: the same step, K1 switched to the pdf-major fp32 kernel (1 warm-up + 2 timed steps)
    fp32_line = None
    if split_form and not args.no_fp32_line:
        for c in ctxs:
            c.set_k1_form("pdf")
        step()
        torch.cuda.synchronize()
        for c in ctxs:
            c.sync()
            c.set_timing(True)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt32 = time.perf_counter() - t0
        k32 = {}
        for c in ctxs:
            for name, ms in c.timings():
                k32[name] = k32.get(name, 0.0) + ms
            c.set_timing(False)
            c.set_k1_form(k1_form)
        if dist_on:
            t32 = torch.tensor([dt32], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t32, op=dist.ReduceOp.MAX)
            dt32 = float(t32[0])
        k1_32 = k32.get("k1_loglikes", 0.0) / (2 * nb)
        fp32_line = {"k1": "fp32 MFMA (v_mfma_f32_16x16x4_f32), pdf-major", "steps": 2, "ms_per_step": dt32 / 2 * 1e3,
                     "value": frames_total * 2 / dt32, "k1_kernel_ms": k1_32,
                     "roofline_frac": k1_flops_per_launch / (k1_32 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS if k1_32 else None}
        ar_events[:] = ar_events[: args.steps]

    return fp32_line


def recipe_and_flat_start_lines(B):
    """recipe_beam_line (beam 6 / retry 40 on the benchmark's own model) and flat_start_line (the same beams with a MISMATCHED model:
    the order-faithful decoders' regime).  -> (recipe_line, flat_line)"""
    import ctypes as C
    import torch
    from kaldi_hmm_gmm_amd import DeviceModel, _lib, synth
    args, ctxs, sets, dm, tm, accs = B.args, B.ctxs, B.sets, B.dm, B.tm, B.accs
    dist_on, dist, dev, backend, n_local, frames_total, model, gc, D = B.dist_on, B.dist, B.dev, B.backend, B.n_local, B.frames_total, B.model, B.gc, B.D
    # the recipe's own beams (egs/yesno/train.py:165-167: beam 6, retry 40) on the same set: no band (a narrow beam fails many
    # certificates: khg_loglikes_reachable), two timed steps, then who was retried / went through the order-faithful decoder
    recipe_line = None
    flat_line = None
    if not args.no_recipe_beam_line and args.beam != 6.0:
        nside = max(1, int(os.environ.get("KHG_BENCH_SIDE_STEPS", "2")))      # timed steps of the two side lines (2; more for an A/B of their difference)

        def recipe_steps(dmodel):
            """`nside` timed steps at beam 6 / retry 40 with `dmodel` -> (seconds, kernel ms, [retried, fallback, failed] utterances over all ranks)."""
            def step_recipe(download=False):
                accs.zero()
                dmodel.invalidate()
                st_all = []
                for s_ in sets:
                    s_.loglikes(dmodel, reachable_only=True, band=False)
                    r_ = s_.align(tm, beam=6.0, retry_beam=40.0, acoustic_scale=0.1, download="summary" if download else False)
                    if download:
                        st_all.append(np.asarray(r_["status"]))
                for s_ in sets:
                    s_.acc_stats(dmodel, tm, accs)
                return st_all
            step_recipe()
            torch.cuda.synchronize()
            for c in ctxs:
                c.sync(); c.set_timing(True)
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nside):
                step_recipe()
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            dtr = time.perf_counter() - t0
            kr = {}
            for c in ctxs:
                for name, ms in c.timings():
                    kr[name] = kr.get(name, 0.0) + ms
                c.set_timing(False)
            st = np.concatenate(step_recipe(download=True)) if n_local else np.zeros(0, np.int32)
            torch.cuda.synchronize()
            cnt = torch.tensor([dtr, float(((st & 2) != 0).sum()), float(((st & 8) != 0).sum()), float(((st & 1) != 0).sum())],
                               device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            if dist_on:
                mx = cnt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
                dtr = float(mx[0])
            return dtr * 2 / nside, {k: v / nside for k, v in sorted(kr.items())}, [int(cnt[1]), int(cnt[2]), int(cnt[3])]

        dtr, kr, cnt = recipe_steps(dm)
        recipe_line = {"beam": 6.0, "retry_beam": 40.0, "k1_cells": "khg_loglikes_reachable (no band)", "steps": 2, "ms_per_step": dtr / 2 * 1e3,
                       "value": frames_total * 2 / dtr, "kernel_ms_per_step": kr,
                       "retried_utts": cnt[0], "fallback_decoder_utts": cnt[1], "failed_utts": cnt[2], "utterances": args.utts,
                       "note": "the recipe's AlignConfig (egs/yesno/train.py:165-167) on the benchmark's set: retried = num_retried of "
                               "decoder-wrappers.cc:68-75, fallback = utterances whose exact-DP beam certificate failed and that the order-"
                               "faithful FasterDecoder kernel decoded; with the TRAINED-like synthetic model of this set (Gaussians ~27 sigma "
                               "apart) the correct path wins by a wide margin, so few utterances leave the certified path -- "
                               "flat_start_line is the other regime"}
        # The regime of a recipe's EARLY realign passes (egs/yesno/train.py:165-202 right after the flat start): the same set scored with a
        # MISMATCHED model (a tenth of the pdfs traded parameters: sure of itself and wrong there, synth.mismatched_model), beam 6 /
        # retry 40.  The best path leaves the beam, the certificate fails and the order-faithful decoders (csrc/faster-decoder.cc:154-335)
        # produce the answer for a large share of the utterances: their time is on this record, beside the headline's.
        dm_flat = None
        try:
            mm = synth.mismatched_model(model, args.flat_fraction, seed=args.seed + 5)
            gc_flat = np.zeros_like(gc)
            _lib.check(_lib.lib.khg_compute_gconsts(mm.num_pdfs, D, _lib.ptr(mm.gauss_off, C.c_int32), _lib.ptr(mm.weights, C.c_float), _lib.ptr(mm.inv_vars, C.c_float),
                                                    _lib.ptr(mm.means_invvars, C.c_float), _lib.ptr(gc_flat, C.c_float), None))
            dm_flat = DeviceModel(ctxs[0], mm.gauss_off, gc_flat, mm.means_invvars, mm.inv_vars)
            dtf, kf, cntf = recipe_steps(dm_flat)
            k2f = kf.get("k2_viterbi_dp", 0.0) + kf.get("k2_viterbi_faithful", 0.0)
            flat_line = {"beam": 6.0, "retry_beam": 40.0, "scoring_model": "synth.mismatched_model(fraction %g): that share of the pdfs traded parameters" % args.flat_fraction, "steps": 2,
                         "ms_per_step": dtf / 2 * 1e3, "value": frames_total * 2 / dtf, "kernel_ms_per_step": kf,
                         "retried_utts": cntf[0], "fallback_decoder_utts": cntf[1], "failed_utts": cntf[2], "utterances": args.utts,
                         "fallback_share": cntf[1] / max(args.utts, 1), "k2_ms_per_step": {"exact_dp": kf.get("k2_viterbi_dp"), "order_faithful": kf.get("k2_viterbi_faithful"), "sum": k2f},
                         "note": "k2_viterbi_faithful runs on a side stream beside K3's first pass (the certified utterances); K3's kernels appear twice "
                                 "per step in kernel_ms_per_step (second pass: the utterances the order-faithful decoders aligned)"}
        except Exception as ex:       # (a side line must not cost the headline)
            flat_line = {"error": repr(ex)}
        finally:
            if dm_flat is not None:
                dm_flat.close()

    return recipe_line, flat_line


