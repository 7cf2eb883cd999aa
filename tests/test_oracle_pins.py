"""Pin the CPU oracle against every known-answer / formula test the reference holds for this path
(SURVEY.md 8c).  The reference's tests re-evaluate closed-form formulas in torch on seeded random
inputs; the same formulas are re-evaluated here in numpy float64 and the oracle must meet the
reference's own tolerances (file:line cited per test)."""
import math

import numpy as np
import pytest

from oracle import oracle as orc


def _rand_gmm(rng, nmix, dim, var_lo=0.05):
    w = rng.random(nmix).astype(np.float32)
    w /= w.sum()
    mean = rng.random((nmix, dim)).astype(np.float32)
    var = (rng.random((nmix, dim)) * (1 - var_lo) + var_lo).astype(np.float32)
    inv_vars = (1 / var).astype(np.float32)
    miv = (mean * inv_vars).astype(np.float32)          # DiagGmm::SetMeans after SetInvVars
    gc, nb = orc.compute_gconsts(w, inv_vars, miv)
    assert nb == 0
    return w, mean, var, inv_vars, miv, gc


def _density_loglikes(w, mean, var, x):
    """python/tests/test_diag_gmm.py:345-347 in float64."""
    w, mean, var, x = (a.astype(np.float64) for a in (w, mean, var, x))
    e = np.exp(((x - mean) ** 2 / (-2 * var)).sum(1)) / np.sqrt((var * math.pi * 2).prod(1))
    return np.log(w * e)


@pytest.mark.parametrize("seed", range(5))
def test_gconsts_formula(seed):
    # python/tests/test_diag_gmm.py:45-51 (torch.allclose defaults: rtol 1e-5, atol 1e-8)
    rng = np.random.default_rng(20230414 + seed)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 8)
    w64, mean64, var64 = w.astype(np.float64), mean.astype(np.float64), var.astype(np.float64)
    expected = np.log(w64) - 0.5 * (8 * math.log(2 * math.pi) + np.log(var64).sum(1) + (mean64 ** 2 / var64).sum(1))
    np.testing.assert_allclose(gc, expected, rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("seed", range(5))
def test_log_likelihood_vs_gaussian_density(seed):
    # python/tests/test_diag_gmm.py:327-349: abs(log_likes - expected) < 1e-4
    rng = np.random.default_rng(100 + seed)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 8)
    x = rng.random(8).astype(np.float32)
    m = orc.OModel([0, 10], gc, miv, inv_vars)
    post, ll = orc.component_posteriors(m, 0, x)
    per = _density_loglikes(w, mean, var, x)
    expected = np.log(np.exp(per).sum())
    assert abs(ll - expected) < 1e-4
    # :351-373 per component, :528-553 posteriors == softmax(loglikes) and logsumexp within 1e-4
    got = orc.loglikes(gc, miv, inv_vars, x)
    np.testing.assert_allclose(got, per, rtol=1e-5, atol=1e-4)
    sm = np.exp(per - per.max()); sm /= sm.sum()
    np.testing.assert_allclose(post, sm, rtol=1e-4, atol=1e-6)
    assert abs(orc.logsumexp(got) - ll) < 1e-5


def test_log_likelihoods_matrix_and_preselect():
    # python/tests/test_diag_gmm.py:375-434
    rng = np.random.default_rng(7)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 3)
    X = rng.random((3, 3)).astype(np.float32)
    for i in range(3):
        got = orc.loglikes(gc, miv, inv_vars, X[i])
        np.testing.assert_allclose(got, _density_loglikes(w, mean, var, X[i]), rtol=1e-5, atol=1e-4)
        idx = [0, 1, 3, 8, 7, 8, 3, 2]
        np.testing.assert_allclose(got[idx], _density_loglikes(w, mean, var, X[i])[idx], atol=1e-4)


def test_fma_order_restatement_matches():
    # the MFMA-order chain differs from the reference-order sums only by fp32 rounding
    rng = np.random.default_rng(3)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 32, 40)
    x = (rng.standard_normal(40) * 2).astype(np.float32)
    a = orc.loglikes(gc, miv, inv_vars, x)
    b = orc.loglikes(gc, miv, inv_vars, x, fma_order=True)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=2e-3)


def test_accumulate_from_diag_semantics():
    # python/tests/test_mle_diag_gmm.py:165-252: occ += post, mean += post (x) x, var += post (x) x^2,
    # weight scales the posteriors, returned log-like within 1e-5 of logsumexp; dtype float64 (:48-90)
    rng = np.random.default_rng(11)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 8)
    m = orc.OModel([0, 10], gc, miv, inv_vars)
    x = rng.random(8).astype(np.float32)
    post, ll = orc.component_posteriors(m, 0, x)
    for weight in (1.0, 0.25):
        acc = orc.OAccs(10, 8, 2)
        got_ll = orc.acc_stats_ali(m, [0, 0, 0], x[None, :], [1], acc, weight=weight)
        assert acc.occ.dtype == np.float64 and acc.mean_acc.dtype == np.float64
        p = (post * np.float32(weight)).astype(np.float32)
        np.testing.assert_array_equal(acc.occ, p.astype(np.float64))
        np.testing.assert_array_equal(acc.mean_acc, np.outer(p, x).astype(np.float32).astype(np.float64))
        np.testing.assert_array_equal(acc.var_acc, np.outer(p, (x * x)).astype(np.float32).astype(np.float64))
        assert abs(got_ll - ll) < 1e-5
        assert acc.trans_acc[1] == 1.0 and acc.trans_acc.sum() == 1.0   # scripts/test_gmm_acc_stats_ali.py:106
        assert acc.total_frames == pytest.approx(weight)


def test_gmm_update_flags():
    # python/tests/test_gmm_update_flags.py:9-39 + csrc/model-common.cc:72-85
    assert orc.augment_gmm_flags(0x2) == 0x7     # v => m => w
    assert orc.augment_gmm_flags(0x1) == 0x5
    assert orc.augment_gmm_flags(0x0) == 0x4     # empty => w
    assert orc.augment_gmm_flags(0xF) == 0xF     # kGmmAll keeps the transition bit


def test_gconst_edge_cases():
    # csrc/diag-gmm.cc:131-146: zero weight -> -inf kept and counted; NaN -> error
    w = np.array([0.0, 1.0], np.float32)
    iv = np.ones((2, 3), np.float32); miv = np.zeros((2, 3), np.float32)
    gc, nb = orc.compute_gconsts(w, iv, miv)
    assert nb == 1 and gc[0] == -np.inf and np.isfinite(gc[1])
    iv[1, 0] = -1.0
    with pytest.raises(orc.OracleError):
        orc.compute_gconsts(w, iv, miv)


def test_threaded_em_pass_driver_counts_every_utterance():
    """orc_em_pass_mt (bench.py's cpu_baseline, variant B): same per-utterance calls on POSIX threads."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from helpers import build
    m, gc, om, ut, cost = build(12, 3, 6, n_utt=9, seed=3, max_phones=4)
    g = dict(ut.graphs)
    g["weight"] = np.where(g["ilabel"] >= 1, g["weight"] + cost[g["ilabel"]], g["weight"]).astype(np.float32)
    for nthr in (1, 3):
        frames, utts, failed, secs = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, num_threads=nthr, acoustic_scale=0.1)
        assert frames == ut.frame_off[-1] and utts == 9 and failed == 0 and secs > 0
    frames, utts, failed, _ = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, first_utt=4, n_utt=3, num_threads=2, acoustic_scale=0.1)
    assert utts == 3 and frames == ut.frame_off[7] - ut.frame_off[4]
    frames, utts, failed, _ = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, num_threads=2, budget_seconds=0.0, acoustic_scale=0.1)
    assert utts == 0 and frames == 0
