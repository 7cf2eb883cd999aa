"""Early-iteration behaviour: align with a MISMATCHED model (1 Gaussian per pdf at the global mean + noise, like the
flat start of a recipe) at beam 6 / retry 40 and report how many utterances need the serial order-faithful decoder
and what the alignment pass costs."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
P, G, D = 5000, 64, 40
U = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
m = synth.make_model(P, G, D, seed=1)
ut = synth.make_utts(m, U, seed=3)
rng = np.random.default_rng(0)
# flat-start-like model: every pdf = the same broad Gaussian, means perturbed a little
mean = ut.feats.mean(0); var = ut.feats.var(0)
go = np.arange(P + 1, dtype=np.int32)
means = (mean[None, :] + noise * np.sqrt(var)[None, :] * rng.standard_normal((P, D))).astype(np.float32)
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
us.loglikes(dm, reachable_only=True); ctx.sync()
for beam, retry in [(200, 0), (10, 40), (6, 40), (200, 0)]:
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
    t0 = time.time()
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
    dt = time.time() - t0
    st = np.asarray(res["status"])
    print(f"beam {beam}/{retry}: {dt*1e3:.1f} ms for {U} utts  exact_dp={(st & 4 > 0).sum()} fallback={(st & 8 > 0).sum()} retried={(st & 2 > 0).sum()} error={(st & 1).sum()}")
