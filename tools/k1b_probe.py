"""K1-only microbenchmark of the bf16x3 form against its MFMA floor, by utterance length (tile balance):
floor = (32-frame tiles x pdfs x 2 W tiles x 30 MFMAs x 32 cycles) / (1024 SIMDs x 2.4 GHz)."""
import os, sys
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
forms = os.environ.get("FORMS", "bf16x3").split(",")
for T in [int(a) for a in sys.argv[1:]] or [512, 256, 300, 160, 128]:
    U = 6_000_000 // T
    frame_off = (np.arange(U + 1) * T).astype(np.int64)
    feats = rng.standard_normal((U * T, D)).astype(np.float32)
    us = UtteranceSet(ctx, None, frame_off, feats)
    us.set_pdf_list(np.sort(rng.choice(P, 75, replace=False)).astype(np.int32))
    for form in forms:
        ctx.set_k1_form(form)
        us.loglikes(dm)
        try:
            ctx.sync()
        except Exception as ex:
            print("   (", str(ex)[:60], ")")
        ctx.set_timing(True)
        for _ in range(3): us.loglikes(dm)
        ms = np.mean([t for n, t in ctx.timings() if n == "k1_loglikes"])
        ctx.set_timing(False)
        try:
            ctx.sync()
        except Exception:
            pass
        n32 = (T + 31) // 32
        floor_ms = U * n32 * 75 * 2 * 30 * 32 / (1024 * 2.4e9) * 1e3
        fl = U * T * 75 * (4 * D * G + 5 * G)
        print(f"T={T} {form}: {ms:.2f} ms; bf16x3 MFMA floor {floor_ms:.2f} ms -> {floor_ms / ms:.3f}; algorithmic {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
    us.close()
