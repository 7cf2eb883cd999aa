"""The two-piece fp16 representations behind the fp16-matrix-core K1 forms, emulated in numpy on the CPU (no GPU):

  f16x2   (csrc/khg_k1_f16x2.hip.inc, k1h_split2):  v = v1 + v2 2^-11,  v1 = fp16(v),  v2 = fp16((v - v1) 2^11)
  f16x2s  (csrc/khg_k1_f16x2s.hip.inc, k1s_split):   v = v1 + v2,        v1 = fp16(v),  v2 = fp16(v - v1)

Stated bounds (kernel headers, DESIGN.md section 3, bench.py dtype_note):
  * |v - v1| <= 2^-11 |v|  (fp16 carries an 11-bit significand)
  * |v - (v1 + v2 ...)| <= 2^-23 |v| while the residual is a normal fp16 number (f16x2: |v| >= 2^-13, an absolute 2^-36
    below); for f16x2s, whose residual is NOT rescaled, max(2^-23 |v|, 2^-25) -- below |v| = 2^-2 the residual falls on
    fp16's subnormal grid (spacing 2^-24)
  * the three kept products w1 x1 + w1 x2 + w2 x1 miss w x by at most 2^-21 |w x| (+ the absolute floor for f16x2s)
The kernels scale their operands into fp16's range first (feature columns peak in [2^14, 2^15)), so the interesting range
is |v| in [2^-14, 2^15].
"""
import numpy as np


def split_scaled(v):        # k1h_split2
    v = v.astype(np.float32)
    v1 = v.astype(np.float16)
    r = (v - v1.astype(np.float32)) * np.float32(2048.0)
    v2 = r.astype(np.float16)
    return v1, v2


def split_plain(v):         # k1s_split
    v = v.astype(np.float32)
    v1 = v.astype(np.float16)
    v2 = (v - v1.astype(np.float32)).astype(np.float16)
    return v1, v2


def _values(rng, n, lo_exp, hi_exp):
    """n fp32 values with random 24-bit significands, exponents uniform in [lo_exp, hi_exp), both signs; plus range edges."""
    mant = 1.0 + rng.integers(0, 1 << 23, size=n).astype(np.float64) / (1 << 23)
    v = np.ldexp(mant, rng.integers(lo_exp, hi_exp, size=n)) * rng.choice([-1.0, 1.0], size=n)
    edges = []
    for e in (hi_exp - 1, 14, 0, -2, -3, -13, -14):
        base = np.float32(2.0 ** e)
        edges += [base, np.nextafter(base, np.float32(np.inf)), np.nextafter(base, np.float32(0)), base * np.float32(1.9999999),
                  base * np.float32(1.00048828125), base * np.float32(1.000732421875)]
    edges += [np.float32(32767.998), np.float32(32768.0 - 2.0 ** -9)]
    return np.concatenate([v.astype(np.float32), np.array(edges, np.float32), -np.array(edges, np.float32)])


def test_first_piece_is_an_11_bit_rounding():
    rng = np.random.default_rng(0)
    v = _values(rng, 1 << 20, -14, 15)
    v1, _ = split_plain(v)
    rel = np.abs(v.astype(np.float64) - v1.astype(np.float64)) / np.abs(v.astype(np.float64))
    assert rel.max() <= 2.0 ** -11
    assert rel.max() > 2.0 ** -11.1        # and no better than that: NOT 2^-12


def test_f16x2_prescaled_residual_keeps_2_pow_minus_23():
    rng = np.random.default_rng(1)
    v = _values(rng, 1 << 20, -14, 15)      # the kernel's own operand range after its per-k scaling
    v1, v2 = split_scaled(v)
    rec = v1.astype(np.float64) + v2.astype(np.float64) * 2.0 ** -11
    err = np.abs(v.astype(np.float64) - rec)
    rel = err / np.abs(v.astype(np.float64))
    normal = np.abs(v) >= 2.0 ** -13          # both pieces are normal fp16 numbers or the residual's subnormal grid is fine enough
    assert rel[normal].max() <= 2.0 ** -23
    assert rel[normal].max() > 2.0 ** -23.05  # the bound is attained: 23 bits, not 24
    assert (err[~normal] <= 2.0 ** -36).all() and (~normal).any()      # at the bottom of fp16's range: an absolute 2^-25 2^-11
    assert np.isfinite(v2.astype(np.float32)).all()


def test_f16x2s_plain_residual_relative_above_quarter_absolute_below():
    rng = np.random.default_rng(2)
    v = _values(rng, 1 << 20, -14, 15)
    v1, v2 = split_plain(v)
    err = np.abs(v.astype(np.float64) - (v1.astype(np.float64) + v2.astype(np.float64)))
    bound = np.maximum(2.0 ** -23 * np.abs(v.astype(np.float64)), 2.0 ** -25)
    assert (err <= bound).all()
    big = np.abs(v) >= 0.25
    assert (err[big] <= 2.0 ** -23 * np.abs(v[big].astype(np.float64))).all()
    # values below fp16's normal range altogether: still inside the absolute floor
    tiny = _values(rng, 1 << 16, -30, -14)
    t1, t2 = split_plain(tiny)
    assert (np.abs(tiny.astype(np.float64) - (t1.astype(np.float64) + t2.astype(np.float64))) <= 2.0 ** -25).all()


def test_three_products_miss_the_product_by_2_pow_minus_21():
    rng = np.random.default_rng(3)
    n = 1 << 20
    w = _values(rng, n, -2, 15)[:n]
    x = _values(rng, n, -2, 15)[:n]
    exact = w.astype(np.float64) * x.astype(np.float64)
    for split, s2 in ((split_scaled, 2.0 ** -11), (split_plain, 1.0)):
        w1, w2 = (a.astype(np.float64) for a in split(w))
        x1, x2 = (a.astype(np.float64) for a in split(x))
        kept = w1 * x1 + (w1 * x2 + w2 * x1) * s2          # the MFMA's products are exact; sums here in fp64
        rel = np.abs(kept - exact) / np.abs(exact)
        assert rel.max() <= 2.0 ** -21
        assert rel.max() > 2.0 ** -22.5                      # ... and the worst case is approached: "about 22-bit operands"
    # a log-likelihood-like sum of 80 terms: error relative to sum |w x| far inside the tests' 1e-6 B
    K = 80
    wk, xk = w[: (n // K) * K].reshape(-1, K), x[: (n // K) * K].reshape(-1, K)
    w1, w2 = (a.astype(np.float64) for a in split_plain(wk))
    x1, x2 = (a.astype(np.float64) for a in split_plain(xk))
    s = (w1 * x1 + w1 * x2 + w2 * x1).sum(1)
    ex = (wk.astype(np.float64) * xk.astype(np.float64)).sum(1)
    B = np.abs(wk.astype(np.float64) * xk.astype(np.float64)).sum(1)
    assert (np.abs(s - ex) / B).max() < 2.0 ** -21.5
