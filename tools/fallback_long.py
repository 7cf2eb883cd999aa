"""Order-faithful decoder on LONG transcripts (graphs of 1000 .. 3000 states): flat-start-like model, beam 6 / retry 40 so that
a good part of the utterances leaves the certified path; the wave form (LDS tables / graph tables in HBM scratch) against the
one-lane form.  usage: fallback_long.py [n_utt] [min_phones] [max_phones]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
P, G, D = 3000, 1, 40
U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 340
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 500
m = synth.make_model(P, 2, D, seed=1)
ut = synth.make_utts(m, U, seed=3, min_phones=lo, max_phones=hi)
rng = np.random.default_rng(0)
mean = ut.feats.mean(0); var = ut.feats.var(0)
go = np.arange(P + 1, dtype=np.int32)
means = (mean[None, :] + 0.3 * np.sqrt(var)[None, :] * rng.standard_normal((P, D))).astype(np.float32)
iv = np.tile((1.0 / var).astype(np.float32), (P, 1))
miv = (means * iv).astype(np.float32)
w = np.ones(P, np.float32)
gc = np.zeros(P, np.float32)
_lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(go, C.c_int32), _lib.ptr(w, C.c_float), _lib.ptr(iv, C.c_float), _lib.ptr(miv, C.c_float), _lib.ptr(gc, C.c_float), None))
ctx = Context(0)
dm = DeviceModel(ctx, go, gc, miv, iv)
tm = DeviceTransitions(ctx, m.id2pdf)
cost = np.zeros(m.num_tids + 1, np.float32)
_lib.check(_lib.lib.khg_scaled_trans_cost(m.num_tids, _lib.ptr(m.log_probs, C.c_float), _lib.ptr(m.non_self_loop_log_probs, C.c_float), _lib.ptr(m.id2state, C.c_int32), _lib.ptr(m.is_self_loop, C.c_uint8), 1.0, 0.1, _lib.ptr(cost, C.c_float)))
tm.set_trans_cost(cost)
us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
S = np.diff(ut.graphs["state_off"])
print(f"{U} utterances, {lo}..{hi} phones: states {S.min()}..{S.max()}, frames {int(ut.frame_off[-1])}")
us.loglikes(dm, reachable_only=True); ctx.sync()
names = {0: "wave (auto)", 2: "wave, graph in HBM", 1: "one lane"}
for kw in (dict(beam=6.0, retry_beam=40.0), dict(beam=200.0, retry_beam=0.0, max_active=7000), dict(beam=10.0, retry_beam=0.0, max_active=1000)):
    ref = None
    for mode in (0, 2, 1):
        ctx.set_option("k2_serial", mode)
        res = us.align(tm, acoustic_scale=0.1, **kw); ctx.sync()
        ctx.set_timing(True)
        t0 = time.time()
        res = us.align(tm, acoustic_scale=0.1, **kw); ctx.sync()
        dt = time.time() - t0
        km = dict(ctx.timings()); ctx.set_timing(False)
        st = np.asarray(res["status"])
        same = "" if ref is None else f"  identical to the wave form: {bool(np.array_equal(ref['ali'], res['ali']) and np.array_equal(ref['status'], res['status']))}"
        if ref is None:
            ref = res
        print(f"{kw} {names[mode]}: {dt*1e3:.1f} ms  fallback={(st & 8 > 0).sum()} retried={(st & 2 > 0).sum()} error={(st & 1).sum()}  " +
              ", ".join(f"{k} {v:.2f}" for k, v in km.items() if k.startswith("k2")) + same)
ctx.set_option("k2_serial", 0)
