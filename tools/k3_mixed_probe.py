"""K3 on a model with ONE pdf that a split pushed past 64 Gaussians, against the uniform 64-Gaussian model (the form
used to be chosen from the model-wide maximum, so one such pdf put all 5000 on the chunk-per-block form)."""
import sys
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
P, D = 5000, 40
U = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = Context(0)
for name, grown in (("uniform 64", {}), ("pdf 17 at 130 Gaussians", {17: 130}), ("pdfs 17 / 2000 at 130 / 256", {17: 130, 2000: 256})):
    counts = np.full(P, 64); 
    for p, g in grown.items(): counts[p] = g
    m = synth.make_model(P, 64, D, seed=1, gauss_counts=counts)
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
    ctx.set_timing(False)
    print(f"{name}: {int(ut.frame_off[-1])} frames, " + ", ".join(f"{k} {v:.3f} ms" for k, v in sorted(km.items())), flush=True)
    accs.close(); us.close(); tm.close(); dm.close()
