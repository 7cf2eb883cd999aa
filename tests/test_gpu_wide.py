"""Shapes past the tile kernels' limits: feature dimensions 80 < D <= KHG_MAX_DIM run K1 on the vector ALUs
(k1w_loglikes) and K3 in its any-dimension form (k3_accumulate<0>); pdfs of more than 128 Gaussians run the MFMA block form of K3 up to 256 Gaussians at D <= 40
and the VALU K3 otherwise.  The reference has no such limits (csrc/diag-gmm.h:243-256).  Same bounds as tests/test_gpu_parity.py: log-likes
within 1e-5 + 1e-6 B max(1, D / 80) of the fp64 value (a sequential fp32 chain of 2 D terms: the bound of
tests/test_gpu_parity.py was stated for 2 D <= 160 terms and grows with their number; the oracle's fp32 sums must meet it too), alignments identical to the oracle decoder's, statistics to 2e-5, the M-step's
parameters bit-identical to the host form."""
import numpy as np
import pytest

from helpers import build, exact_loglikes, oracle_graph, utt_feats
from kaldi_hmm_gmm_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

LL_ATOL, LL_RTOL = 1e-5, 1e-6


def _device(ctx, m, gc, ut, cost, weights=False):
    from kaldi_hmm_gmm_amd import DeviceModel, DeviceTransitions, UtteranceSet

    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights if weights else None)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    return dm, tm, us


@pytest.mark.parametrize("P,G,D,ragged", [(12, 16, 120, True), (9, 200, 40, True), (5, 256, 39, False), (4, 260, 40, True), (6, 150, 81, False), (5, 3, 257, True)])
def test_wide_loglikes_align_accstats_vs_oracle(ctx, P, G, D, ragged):
    from kaldi_hmm_gmm_amd import DeviceAccs

    m, gc, om, ut, cost = build(P, G, D, n_utt=8, seed=P + D, ragged=ragged, max_phones=4)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    # K1
    us.loglikes(dm)
    got = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    for u in range(us.n_utt):
        pl = pdfs[poff[u]: poff[u + 1]]
        x = utt_feats(ut, u)
        exact, bound = exact_loglikes(m, gc, x, pl)
        tol = LL_ATOL + LL_RTOL * bound * max(1.0, D / 80)
        assert got[u].shape == exact.shape and np.isfinite(got[u]).all()
        assert (np.abs(got[u] - exact) <= tol).all(), f"utt {u}: max err/tol {(np.abs(got[u] - exact) / tol).max()}"
        want = orc.loglikes_matrix(om, x, pl)
        assert (np.abs(want - exact) <= tol).all(), "oracle itself outside the fp32 bound"
        assert (np.abs(got[u] - want) <= 2 * tol).all()
    # K1 + K2 against the oracle decodable + FasterDecoder
    res = us.align(tm, acoustic_scale=0.1)
    for u in range(us.n_utt):
        want = orc.align_utterance(oracle_graph(ut, u, cost), om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1)
        a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
        assert int(res["status"][u]) & 1 == 0 and want["status"] == 0
        assert (a == want["ali"]).all()
        assert res["like"][u] == pytest.approx(want["like"], rel=2e-5)
    # K3 from the reference alignment
    us.upload_ali(ut.ref_ali)
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs, weight=1.0)
    st = accs.download()
    oa = orc.OAccs(int(m.gauss_off[-1]), D, m.num_tids)
    for u in range(us.n_utt):
        orc.acc_stats_ali(om, m.id2pdf, utt_feats(ut, u), ut.ref_ali[ut.frame_off[u]: ut.frame_off[u + 1]], oa)
    assert (st["trans_acc"] == oa.trans_acc).all() and st["total_frames"] == oa.total_frames
    assert st["total_log_like"] == pytest.approx(oa.total_log_like, rel=2e-6)
    np.testing.assert_allclose(st["occ"], oa.occ, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(st["mean_acc"], oa.mean_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.mean_acc).max())
    np.testing.assert_allclose(st["var_acc"], oa.var_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.var_acc).max())
    # twice the same statistics, bit for bit (sorted bucketing + fixed flush order)
    accs2 = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs2, weight=1.0)
    st2 = accs2.download()
    assert np.array_equal(st2["trans_acc"], st["trans_acc"])
    np.testing.assert_allclose(st2["mean_acc"], st["mean_acc"], rtol=1e-12, atol=1e-12 * np.abs(st["mean_acc"]).max())


def test_wide_loglikes_are_the_fp32_chain_bit_for_bit(ctx):
    """One Gaussian per pdf (log-sum-exp of one term = that term up to exp(0) = 1, log(1) = 0): the any-dimension K1 gives the
    k-ordered fmaf chain of the fp32 MFMA forms exactly (oracle/khg_oracle.c:orc_loglikes_fma_order), at D odd and even."""
    for D in (120, 97):
        m, gc, om, ut, cost = build(10, 1, D, n_utt=3, seed=5 + D)
        dm, tm, us = _device(ctx, m, gc, ut, cost)
        us.loglikes(dm)
        got = us.download_loglikes()
        poff, pdfs = us.pdf_lists()
        for u in range(us.n_utt):
            f = utt_feats(ut, u)
            for j, p in enumerate(pdfs[poff[u]: poff[u + 1]]):
                g0 = m.gauss_off[p]
                for t in (0, f.shape[0] // 2, f.shape[0] - 1):
                    v = orc.loglikes(gc[g0:g0 + 1], m.means_invvars[g0:g0 + 1], m.inv_vars[g0:g0 + 1], f[t], fma_order=True)
                    assert got[u][j, t] == v[0]


def test_wide_em_iteration_on_the_device(ctx):
    """align -> acc-stats -> device M-step -> split -> align again at D = 120: the re-estimated parameters equal the host
    M-step's on the same statistics bit for bit, and the new model (re-laid-out, no tile image) scores and aligns."""
    from kaldi_hmm_gmm_amd import DeviceAccs, MleDiagGmmOptions
    from kaldi_hmm_gmm_amd import mle as khg_mle

    P, G, D = 10, 6, 120
    m, gc, om, ut, cost = build(P, G, D, n_utt=10, seed=77, ragged=True, max_phones=4)
    dm, tm, us = _device(ctx, m, gc, ut, cost, weights=True)
    us.loglikes(dm)
    res = us.align(tm, acoustic_scale=0.1)
    assert (res["status"] & 1 == 0).all()
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs, weight=1.0)
    st = accs.download()
    opts = MleDiagGmmOptions(min_gaussian_occupancy=3)
    h = khg_mle._flat_update(opts, m.gauss_off, st["occ"], st["mean_acc"], st["var_acc"], 0x7, 0x7, m.weights, m.means_invvars,
                             m.inv_vars)
    r = dm.mle_update(accs, opts, 0x7)
    d = dm.download()
    assert np.array_equal(d["gauss_off"], h[0]) and r["removed"] == h[9]
    for name, want in (("weights", h[1]), ("means_invvars", h[3]), ("inv_vars", h[4])):
        np.testing.assert_array_equal(d[name], want, err_msg=name)
    accs.relayout(dm)
    cur = np.diff(d["gauss_off"])
    tgt = (cur + 1).astype(np.int32)
    dm.split(tgt, 0.01, np.random.default_rng(3).standard_normal((P, D)).astype(np.float32))
    accs.relayout(dm)
    us.loglikes(dm)
    res2 = us.align(tm, acoustic_scale=0.1)
    assert (res2["status"] & 1 == 0).all()
    us.acc_stats(dm, tm, accs, weight=1.0)
    st2 = accs.download()
    assert st2["total_frames"] == ut.frame_off[-1] and st2["occ"].shape[0] == int(tgt.sum())
    # ... and its scores are those of the downloaded parameters
    from types import SimpleNamespace
    d2 = dm.download()
    m2 = SimpleNamespace(gauss_off=d2["gauss_off"], means_invvars=d2["means_invvars"], inv_vars=d2["inv_vars"])
    got = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    for u in range(us.n_utt):
        pl = pdfs[poff[u]: poff[u + 1]]
        exact, bound = exact_loglikes(m2, d2["gconsts"], utt_feats(ut, u), pl)
        assert (np.abs(got[u] - exact) <= LL_ATOL + LL_RTOL * bound * D / 80).all()


def test_dimension_limit_is_an_error(ctx):
    from kaldi_hmm_gmm_amd import DeviceModel, KhgError

    D = 513
    go = np.array([0, 1, 2], np.int32)
    with pytest.raises(KhgError, match="feature dim > 512"):
        DeviceModel(ctx, go, np.zeros(2, np.float32), np.zeros((2, D), np.float32), np.ones((2, D), np.float32))


def test_wide_loglikes_tolerate_an_infinite_first_gconst(ctx):
    """A zero-weight FIRST component (gconst = log 0 = -inf, which ComputeGconsts keeps: csrc/diag-gmm.cc:132-146) must not poison
    the online log-sum-exp of the D > 80 kernel: the pdf's log-likelihood is the log-sum over the other components."""
    from kaldi_hmm_gmm_amd import DeviceModel, UtteranceSet

    P, G, D = 3, 5, 96
    m, gc, om, ut, cost = build(P, G, D, n_utt=2, seed=91)
    gc = gc.copy()
    gc[int(m.gauss_off[1])] = -np.inf              # pdf 1: first component dead
    gc[int(m.gauss_off[2]) + 2] = -np.inf          # pdf 2: a middle one
    frame_off = np.array([0, 70], np.int64)
    feats = (np.random.default_rng(3).standard_normal((70, D)) * 1.5).astype(np.float32)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    us = UtteranceSet(ctx, None, frame_off, feats)
    pl = np.arange(P, dtype=np.int32)
    us.set_pdf_list(pl)
    us.loglikes(dm)
    got = us.download_loglikes()[0]
    assert np.isfinite(got).all()
    x = feats.astype(np.float64)
    for p in range(P):
        a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        live = np.isfinite(gc[a:b])
        miv, iv, g = (v[a:b][live].astype(np.float64) for v in (m.means_invvars, m.inv_vars, gc))
        ll = g[None, :] + x @ miv.T - 0.5 * (x * x) @ iv.T
        exact = np.logaddexp.reduce(ll, axis=1)
        bound = (np.abs(g)[None, :] + np.abs(x) @ np.abs(miv).T + 0.5 * (x * x) @ iv.T).max(1)
        assert (np.abs(got[p] - exact) <= LL_ATOL + LL_RTOL * bound * 2).all(), p
    us.close()
