"""K3 alone at BASELINE config #5's shape (10 000 pdfs x 128 Gaussians, D = 80): k3_accumulate_block16 per pass.  With KHG_LIBRARY /
LD_LIBRARY_PATH pointing at a knock-out build (-DK3B16_KO=n, results wrong) it prices one part of the kernel."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
P, G, D = 10000, 128, 80
U = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = Context(0)
m = synth.make_model(P, G, D, seed=1)
ut = synth.make_utts(m, U, seed=3)
gc = np.zeros(m.weights.shape[0], np.float32)
_lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float), _lib.ptr(m.inv_vars, C.c_float), _lib.ptr(m.means_invvars, C.c_float), _lib.ptr(gc, C.c_float), None))
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf)
us = UtteranceSet(ctx, None, ut.frame_off, ut.feats)
us.upload_ali(ut.ref_ali)
accs = DeviceAccs(ctx, dm, tm)
for _ in range(2):
    accs.zero(); us.acc_stats(dm, tm, accs)
ctx.sync(); ctx.set_timing(True)
for _ in range(3):
    accs.zero(); us.acc_stats(dm, tm, accs)
ctx.sync()
km = {}
for k, v in ctx.timings(): km[k] = km.get(k, 0.0) + v / 3
print(os.environ.get("KHG_LIBRARY", "product"), f"{int(ut.frame_off[-1])} frames:", ", ".join(f"{k} {v:.3f} ms" for k, v in sorted(km.items())), flush=True)
