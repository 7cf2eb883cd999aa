"""K1-only microbenchmark: algorithmic TFLOP/s as a function of utterance length (tile balance)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from kaldi_hmm_gmm_amd import Context, DeviceModel, UtteranceSet, synth, _lib
import ctypes as C
P, G, D = 5000, 64, 40
m = synth.make_model(P, G, D, seed=1)
gc = np.zeros(m.weights.shape[0], np.float32)
_lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float), _lib.ptr(m.inv_vars, C.c_float), _lib.ptr(m.means_invvars, C.c_float), _lib.ptr(gc, C.c_float), None))
ctx = Context(0)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
rng = np.random.default_rng(0)
for T in [int(a) for a in sys.argv[1:]] or [300, 384, 320, 256, 288]:
    U = 3_000_000 // T
    frame_off = (np.arange(U + 1) * T).astype(np.int64)
    feats = rng.standard_normal((U * T, D)).astype(np.float32)
    us = UtteranceSet(ctx, None, frame_off, feats)
    # per-utterance random pdf lists are not supported for features-only sets: use one random list of 75
    import os
    us.set_pdf_list(np.full(75, 7, np.int32) if os.environ.get('SAMEPDF') else np.sort(rng.choice(P, 75, replace=False)).astype(np.int32))
    us.loglikes(dm); ctx.sync()
    ctx.set_timing(True)
    for _ in range(3): us.loglikes(dm)
    ms = np.mean([t for n, t in ctx.timings()])
    ctx.set_timing(False)
    fl = U * T * 75 * (4 * D * G + 5 * G)
    print(f"T={T}: {ms:.2f} ms, {fl / ms / 1e9:.1f} TFLOP/s algorithmic ({fl / ms / 1e9 / 157.3:.3f} of peak)")
    us.close()
