import sys, time
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
P, G, D = 5000, 64, 40
m = synth.make_model(P, G, D, seed=1)
gc = np.zeros(m.weights.shape[0], np.float32)
_lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float), _lib.ptr(m.inv_vars, C.c_float), _lib.ptr(m.means_invvars, C.c_float), _lib.ptr(gc, C.c_float), None))
U = 100000
ut = synth.make_utts(m, U, seed=3)
ctx = Context(0)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf)
tm.set_trans_cost(np.zeros(m.num_tids + 1, np.float32))
t0 = time.time(); us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs); ctx.sync(); t1 = time.time()
us.loglikes(dm, reachable_only=True); ctx.sync(); t2 = time.time()
us.loglikes(dm, reachable_only=True); ctx.sync(); t3 = time.time()
print(f"utts_create {t1-t0:.2f}s  first loglikes (plan + pack + K1) {t2-t1:.2f}s  second {t3-t2:.3f}s")
