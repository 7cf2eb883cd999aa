"""VERDICT r5 item 6, priced before it is built: how often would a WAVE-UNIFORM dominated-tile shortcut of K1's log-sum-exp fire?
The shortcut skips the 16 v_exp_f32 of a chain (32 Gaussians x 32 frames of one pdf tile) when, for every one of the 32 frames, the
largest per-Gaussian log-likelihood of the tile exceeds the second largest by more than 24 (the sum then differs from the maximum by
< 32 e^-24).  Evaluated in float64 on the benchmark's own synthetic law (5000 x 64 x 40, means 3 N(0, 1),
variances U[0.5, 2]) for the cells K1 computes: a frame against the pdfs on its utterance's graph (one of them its own)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from kaldi_hmm_gmm_amd import synth
P, G, D = 600, 64, 40                # the law does not depend on P; 600 pdfs keep this a few seconds
for name, ms in (("trained-like (mean_scale 3.0)", 3.0), ("overlapping Gaussians (mean_scale 0.3)", 0.3)):
    m = synth.make_model(P, G, D, seed=1, mean_scale=ms)
    ut = synth.make_utts(m, 40, seed=3)
    rng = np.random.default_rng(0)
    x = ut.feats.astype(np.float64)
    N = x.shape[0]
    gc = np.log(m.weights.astype(np.float64)) - 0.5 * D * np.log(2 * np.pi) + (0.5 * np.log(m.inv_vars.astype(np.float64)) - 0.5 * m.means_invvars.astype(np.float64) ** 2 / m.inv_vars).sum(1)
    fire_own = fire_other = tiles_own = tiles_other = 0
    lane_own = lane_other = 0.0
    for u in range(40):
        f0, f1 = int(ut.frame_off[u]), int(ut.frame_off[u + 1])
        pdfs = np.unique(ut.frame_pdf[f0:f1])
        for t0 in range(f0, f1 - 31, 32):          # 32-frame tiles
            xs = x[t0:t0 + 32]
            own = np.bincount(ut.frame_pdf[t0:t0 + 32]).argmax()
            for p in pdfs[:12]:
                for half in (0, 1):                # a chain = one 32-Gaussian tile of the pdf
                    a = m.gauss_off[p] + 32 * half
                    ll = gc[a:a + 32][None, :] + xs @ m.means_invvars[a:a + 32].astype(np.float64).T - 0.5 * (xs * xs) @ m.inv_vars[a:a + 32].astype(np.float64).T
                    srt = np.sort(ll, axis=1)
                    gap = srt[:, -1] - srt[:, -2]
                    ok = gap > 24.0
                    if p == own:
                        tiles_own += 1; fire_own += bool(ok.all()); lane_own += ok.mean()
                    else:
                        tiles_other += 1; fire_other += bool(ok.all()); lane_other += ok.mean()
    print(f"{name}: chains of the tile's own pdf: {fire_own}/{tiles_own} fire (frames with gap > 24: {lane_own / max(tiles_own, 1):.3f}); "
          f"chains of the other pdfs on the graph: {fire_other}/{tiles_other} fire (frames with gap > 24: {lane_other / max(tiles_other, 1):.3f})")
