"""How many utterances of the bench workload leave the exact-DP fast path (status bit 8 = fallback)?"""
import sys
sys.path.insert(0, '.')
import numpy as np, ctypes as C, time
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
from kaldi_hmm_gmm_amd.align import add_transition_probs
P, G, D = 5000, 64, 40
U = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = synth.make_model(P, G, D, seed=1)
gc = np.zeros(m.weights.shape[0], np.float32)
_lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float), _lib.ptr(m.inv_vars, C.c_float), _lib.ptr(m.means_invvars, C.c_float), _lib.ptr(gc, C.c_float), None))
ut = synth.make_utts(m, U, seed=3)
ctx = Context(0)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf)
cost = np.zeros(m.num_tids + 1, np.float32)
_lib.check(_lib.lib.khg_scaled_trans_cost(m.num_tids, _lib.ptr(m.log_probs, C.c_float), _lib.ptr(m.non_self_loop_log_probs, C.c_float), _lib.ptr(m.id2state, C.c_int32), _lib.ptr(m.is_self_loop, C.c_uint8), 1.0, 0.1, _lib.ptr(cost, C.c_float)))
tm.set_trans_cost(cost)
us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
us.loglikes(dm)
for beam, retry in [(200, 0), (10, 40), (6, 20)]:
    t0 = time.time()
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1)
    st = np.asarray(res["status"])
    print(f"beam {beam}/{retry}: {time.time()-t0:.3f}s  n={len(st)} exact_dp={(st & 4 > 0).sum()} fallback={(st & 8 > 0).sum()} retried={(st & 2 > 0).sum()} error={(st & 1).sum()}")
