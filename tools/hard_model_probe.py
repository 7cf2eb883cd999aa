"""Which synthetic model puts a useful share of utterances outside the beam certificate at the recipe's beams (6 / retry 40)?
Sweeps the share of pdfs that traded parameters (synth.mismatched_model) at the benchmark's model shape and prints, per
spread, how many utterances the exact DP certifies, how many go through the order-faithful decoder, how many are retried /
fail, and what the K2 kernels cost."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
P, G, D = 5000, 64, 40
U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scales = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.01, 0.03, 0.1, 0.3]
ctx = Context(0)
for ms in scales:
    m0 = synth.make_model(P, G, D, seed=1)
    ut = synth.make_utts(m0, U, seed=3)
    m = synth.mismatched_model(m0, ms, seed=2)
    gc = np.zeros(m.weights.shape[0], np.float32)
    _lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float), _lib.ptr(m.inv_vars, C.c_float), _lib.ptr(m.means_invvars, C.c_float), _lib.ptr(gc, C.c_float), None))
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    cost = np.zeros(m.num_tids + 1, np.float32)
    _lib.check(_lib.lib.khg_scaled_trans_cost(m.num_tids, _lib.ptr(m.log_probs, C.c_float), _lib.ptr(m.non_self_loop_log_probs, C.c_float), _lib.ptr(m.id2state, C.c_int32), _lib.ptr(m.is_self_loop, C.c_uint8), 1.0, 0.1, _lib.ptr(cost, C.c_float)))
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    us.loglikes(dm, reachable_only=True); ctx.sync()
    for beam, retry in [(6, 40), (10, 40), (200, 0)]:
        res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
        ctx.set_timing(True)
        t0 = time.time()
        res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
        dt = time.time() - t0
        km = dict(ctx.timings()); ctx.set_timing(False)
        st = np.asarray(res["status"])
        same = float((res["ali"] == ut.ref_ali).mean())
        print(f"mismatch {ms}: beam {beam}/{retry}: {dt*1e3:.1f} ms for {U} utts  exact_dp={(st & 4 > 0).sum()} fallback={(st & 8 > 0).sum()} retried={(st & 2 > 0).sum()} error={(st & 1).sum()} frames=generating path {same:.3f}  kernels " +
              ", ".join(f"{k} {v:.2f}" for k, v in km.items() if k.startswith("k2")), flush=True)
    if len(sys.argv) > 3:      # per-phase cycle stamps of the DP and the chain decoder (stderr)
        ctx.set_option("k2_prof", 1)
        us.align(tm, beam=6, retry_beam=40, acoustic_scale=0.1); ctx.sync()
        ctx.set_option("k2_prof", 0)
    us.close(); tm.close(); dm.close()
