"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (north_star: "integer alignments bit-exact, accumulated stats and log-likelihoods
within a stated fp32 tolerance"):
  * log-likes: |gpu - exact64| <= 1e-5 + 1e-6*B, B = max_g(|gconst| + sum|M x| + 0.5 sum|V x^2|):
    fp32 rounding of a 2D-term sum scales with the magnitude of its terms, not of the result;
    at D=40 this is ~3e-4, the same order as the 1e-4 the reference's own tests accept
    (python/tests/test_diag_gmm.py:342-349).  The oracle (sequential fp32 sums) must meet the
    same bound, and GPU vs oracle twice it.
  * alignments: bit-exact when both sides read the SAME score matrix
  * statistics: rtol 2e-5 on occ / mean / var accumulators, transition counts exact
"""
import numpy as np
import pytest

from helpers import build, exact_loglikes, oracle_graph, utt_feats
from kaldi_hmm_gmm_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

LL_ATOL, LL_RTOL = 1e-5, 1e-6


def _device(ctx, m, gc, ut, cost):
    from kaldi_hmm_gmm_amd import DeviceModel, DeviceTransitions, UtteranceSet

    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    return dm, tm, us


@pytest.mark.parametrize("P,G,D,ragged", [(30, 8, 39, False), (60, 32, 40, True), (30, 64, 40, False),
                                          (24, 20, 13, True), (12, 128, 80, False),
                                          # small pdfs: the default form packs 4 (<= 8 Gaussians) / 2 (<= 16) pdfs into one MFMA tile
                                          (31, 5, 40, True), (22, 12, 23, True), (9, 16, 40, False), (3, 8, 13, False)])
def test_loglikes_vs_oracle(ctx, P, G, D, ragged, k1_form):
    m, gc, om, ut, cost = build(P, G, D, n_utt=6, seed=P + G, ragged=ragged, max_phones=5)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm)
    got = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    worst = 0.0
    for u in range(us.n_utt):
        pl = pdfs[poff[u]: poff[u + 1]]
        want = orc.loglikes_matrix(om, utt_feats(ut, u), pl)
        assert got[u].shape == want.shape
        exact, bound = exact_loglikes(m, gc, utt_feats(ut, u), pl)
        tol = LL_ATOL + LL_RTOL * bound
        err = np.abs(got[u] - exact)
        worst = max(worst, float((err / tol).max()))
        assert (err <= tol).all(), f"utt {u}: gpu vs exact max err {err.max()}"
        assert (np.abs(want - exact) <= tol).all(), "oracle itself outside the fp32 bound"
        assert (np.abs(got[u] - want) <= 2 * tol).all()
    print("worst err/tol", worst)

@pytest.mark.parametrize("k1", ["f16x2s", "f16x2", "pdf", "utt"])
@pytest.mark.parametrize("P,G,D", [(30, 64, 40), (24, 20, 13), (12, 128, 80)])
def test_loglikes_overlapping_gaussians(ctx, P, G, D, k1, opt):
    """Same bound with the means pulled together (many components contribute to every log-sum-exp instead of one)."""
    import dataclasses

    opt.k1(k1)
    m, _, _, ut, cost = build(P, G, D, n_utt=6, seed=P + G + 1, max_phones=5)
    means = (0.12 * m.means).astype(np.float32)
    m2 = dataclasses.replace(m, means=means, means_invvars=(means * m.inv_vars).astype(np.float32))
    gc = orc.model_gconsts(m2.gauss_off, m2.weights, m2.inv_vars, m2.means_invvars)
    om = orc.OModel(m2.gauss_off, gc, m2.means_invvars, m2.inv_vars)
    feats = (0.12 * ut.feats + np.random.default_rng(4).standard_normal(ut.feats.shape)).astype(np.float32)
    ut2 = dataclasses.replace(ut, feats=feats)
    dm, tm, us = _device(ctx, m2, gc, ut2, cost)
    us.loglikes(dm)
    got = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    spread = []
    for u in range(us.n_utt):
        pl = pdfs[poff[u]: poff[u + 1]]
        x = utt_feats(ut2, u)
        exact, bound = exact_loglikes(m2, gc, x, pl)
        tol = LL_ATOL + LL_RTOL * bound
        assert (np.abs(got[u] - exact) <= tol).all()
        assert (np.abs(got[u] - orc.loglikes_matrix(om, x, pl)) <= 2 * tol).all()
        # log-sum-exp minus the best single component: > 0.5 nats when several components matter
        p0 = int(pl[0]); a, b = int(m2.gauss_off[p0]), int(m2.gauss_off[p0 + 1])
        xs = x.astype(np.float64)
        comp = gc[a:b].astype(np.float64)[None] + xs @ m2.means_invvars[a:b].astype(np.float64).T - 0.5 * (xs * xs) @ m2.inv_vars[a:b].astype(np.float64).T
        spread.append(float((exact[0] - comp.max(1)).mean()))
    assert np.mean(spread) > 0.5


@pytest.fixture(params=["f16x2s", "f16x2", "pdf", "utt"])
def k1_form(request, opt):
    """Every K1 form: f16x2s (the default: fp16 matrix cores, 3 partial products into one accumulator, feature tiles in LDS and
    pdfs dealt to waves), f16x2 (the same products, two accumulators, frame tiles dealt to waves) and the two fp32-MFMA tilings, pdf-major and utterance-major."""
    opt.k1(request.param)
    return request.param


def _check_ll(us, dm, m, gc, ut):
    us.loglikes(dm)
    return _check_ll_only(us, m, gc, ut)


def _check_ll_only(us, m, gc, ut):
    got = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    worst = 0.0
    for u in range(us.n_utt):
        pl = pdfs[poff[u]: poff[u + 1]]
        exact, bound = exact_loglikes(m, gc, utt_feats(ut, u), pl)
        tol = LL_ATOL + LL_RTOL * bound
        assert np.isfinite(got[u]).all()
        err = np.abs(got[u] - exact)
        assert (err <= tol).all(), f"utt {u}: max err/tol {(err / tol).max()}"
        worst = max(worst, float((err / tol).max()))
    return worst


def _rescaled(m, ut, s):
    """The same model and data in other units: feature dimension d multiplied by s[d] (means too, variances by s[d]^2)."""
    import dataclasses

    s = s.astype(np.float32)
    means = (m.means * s).astype(np.float32)
    inv_vars = (m.inv_vars / (s * s)).astype(np.float32)
    m2 = dataclasses.replace(m, means=means, inv_vars=inv_vars, means_invvars=(means * inv_vars).astype(np.float32))
    gc = orc.model_gconsts(m2.gauss_off, m2.weights, m2.inv_vars, m2.means_invvars)
    return m2, gc, dataclasses.replace(ut, feats=(ut.feats * s).astype(np.float32))


@pytest.mark.parametrize("D,G", [(40, 64), (80, 32)])
def test_loglikes_wide_dynamic_range(ctx, D, G, k1_form):
    """Feature dimensions in wildly different units (1e-3 .. 1e3: inverse variances from 1e-6 to 1e6, means x inverse
    variances likewise): the fp32 bound holds in every form.  For f16x2 this is the per-k power-of-two scaling at work --
    unscaled, half of these operands would overflow or vanish in fp16."""
    m, _, _, ut, cost = build(16, G, D, n_utt=5, seed=D + G, max_phones=5)
    s = 10.0 ** np.random.default_rng(D).uniform(-3, 3, D)
    m2, gc, ut2 = _rescaled(m, ut, s)
    assert m2.inv_vars.max() / m2.inv_vars.min() > 1e9
    dm, tm, us = _device(ctx, m2, gc, ut2, cost)
    _check_ll(us, dm, m2, gc, ut2)


def test_loglikes_f16x2_rescale_and_fallback(ctx, opt):
    """The f16x2 form keeps the set's feature planes while the next model fits their scales, re-packs them when it does
    not, and hands over to the fp32-MFMA form (fp32's exponent range) when no scaling fits fp16."""
    import dataclasses
    from kaldi_hmm_gmm_amd import DeviceModel

    opt.k1("f16x2")
    m, gc, om, ut, cost = build(16, 64, 40, n_utt=5, seed=3, max_phones=5)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    ctx.set_timing(True)
    ctx.timings()

    def kernels():
        ctx.sync()
        return [n for n, _ in ctx.timings()]

    _check_ll(us, dm, m, gc, ut)
    k = kernels()
    assert "k1h_pack_x" in k and "k0h_pack_tiles" in k
    # a slightly different model: same planes, new W image
    m_b = dataclasses.replace(m, means_invvars=(m.means_invvars * 1.25).astype(np.float32))
    gc_b = orc.model_gconsts(m_b.gauss_off, m_b.weights, m_b.inv_vars, m_b.means_invvars)
    dm_b = DeviceModel(ctx, m_b.gauss_off, gc_b, m_b.means_invvars, m_b.inv_vars)
    _check_ll(us, dm_b, m_b, gc_b, ut)
    k = kernels()
    assert "k1h_pack_x" not in k and "k0h_pack_tiles" in k
    # the first model again: its image was packed with the same scales and is still valid
    _check_ll(us, dm, m, gc, ut)
    assert "k0h_pack_tiles" not in kernels()
    # dimension 5 in units 1e3 x smaller for the MODEL only (inverse variances x 1e6): no longer fits -> planes re-packed
    s = np.ones(40); s[5] = 1e-3
    m_c, gc_c, _ = _rescaled(m, ut, s)
    dm_c = DeviceModel(ctx, m_c.gauss_off, gc_c, m_c.means_invvars, m_c.inv_vars)
    _check_ll(us, dm_c, m_c, gc_c, ut)
    k = kernels()
    assert "k1h_pack_x" in k and "k0h_pack_tiles" in k
    # inverse variance 1e12 against features of order 1: |w x^2| ~ 1e13 > 2^30 -> the fp32-MFMA form takes over, still inside the bound
    s[5] = 1e-6
    m_d, gc_d, _ = _rescaled(m, ut, s)
    dm_d = DeviceModel(ctx, m_d.gauss_off, gc_d, m_d.means_invvars, m_d.inv_vars)
    _check_ll(us, dm_d, m_d, gc_d, ut)
    k = kernels()
    assert "k1h_pack_x" not in k and "k0h_pack_tiles" not in k and "k1_loglikes" in k
    ctx.set_timing(False)


def test_loglikes_f16x2s_planes_packed_once_and_fallback(ctx, opt):
    """The default form (f16x2s) packs the set's feature planes ONCE -- their exponents depend on the features alone --, re-packs
    only the W image for a new model, and hands over to the two-accumulator f16x2 form when the absolute part of its error bound
    (residual pieces on fp16's subnormal grid) would exceed 2e-6 for the model at hand; results stay inside the fp32 bound."""
    import dataclasses
    from kaldi_hmm_gmm_amd import DeviceModel

    opt.k1("auto")
    m, gc, om, ut, cost = build(16, 64, 40, n_utt=5, seed=3, max_phones=5)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    ctx.set_timing(True)
    ctx.timings()

    def kernels():
        ctx.sync()
        return [n for n, _ in ctx.timings()]

    _check_ll(us, dm, m, gc, ut)
    k = kernels()
    assert "k1s_pack_x" in k and "k0s_pack_tiles" in k and "k1h_pack_x" not in k
    m_b = dataclasses.replace(m, means_invvars=(m.means_invvars * 1.25).astype(np.float32))
    gc_b = orc.model_gconsts(m_b.gauss_off, m_b.weights, m_b.inv_vars, m_b.means_invvars)
    dm_b = DeviceModel(ctx, m_b.gauss_off, gc_b, m_b.means_invvars, m_b.inv_vars)
    _check_ll(us, dm_b, m_b, gc_b, ut)
    k = kernels()
    assert "k1s_pack_x" not in k and "k0s_pack_tiles" in k
    _check_ll(us, dm, m, gc, ut)           # the first model again: its image is still valid
    k = kernels()
    assert "k0s_pack_tiles" not in k and "k1s_pack_x" not in k
    # inverse variances x 1e6 in one dimension: the largest weight column then pushes the common scale S down and the floor of the
    # bound up -> the two-accumulator form (pre-scaled residuals) runs instead
    s = np.ones(40); s[5] = 1e-3
    m_c, gc_c, _ = _rescaled(m, ut, s)
    dm_c = DeviceModel(ctx, m_c.gauss_off, gc_c, m_c.means_invvars, m_c.inv_vars)
    _check_ll(us, dm_c, m_c, gc_c, ut)
    k = kernels()
    assert "k1h_pack_x" in k and "k0h_pack_tiles" in k and "k0s_pack_tiles" not in k
    ctx.set_timing(False)


def test_two_contexts_on_two_streams_share_one_model(ctx):
    """Two utterance sets with DIFFERENT scale exponents (features in other units) on two contexts / streams, one model: each
    set's K1 needs its own fp16 W image of that model; the lazily re-packed image must never be read half-written or overwritten
    under a running K1 on the other stream (the packs and the reads are ordered by events inside the library)."""
    from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet

    import dataclasses

    m, gc, om, ut, cost = build(40, 64, 40, n_utt=24, seed=5, max_phones=8)
    ctx2 = Context(0)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    us_a = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    # set B: the SAME model on features stretched by 4 / shrunk by 4 in alternating dimensions -> other feature exponents
    sc = np.ones(40, np.float32); sc[::2] = 4.0; sc[1::2] = 0.25
    ut_c = dataclasses.replace(ut, feats=(ut.feats * sc).astype(np.float32))
    us_b = UtteranceSet(ctx2, tm, ut_c.frame_off, ut_c.feats, graphs=ut.graphs)
    for form in ("f16x2s", "f16x2"):
        ctx.set_k1_form(form); ctx2.set_k1_form(form)
        for c in (ctx, ctx2):
            c.sync(); c.set_timing(True); c.timings()
        for _ in range(6):                              # alternate without synchronising: packs and reads interleave on the two streams
            us_a.loglikes(dm)
            us_b.loglikes(dm)
        ctx.sync(); ctx2.sync()
        names = [n for c in (ctx, ctx2) for n, _ in c.timings()]
        for c in (ctx, ctx2):
            c.set_timing(False)
        if form == "f16x2s":
            # the model settles on the element-wise minimum of the two sets' feature exponents: one image for both after the
            # first round, no re-pack of the 100 MB-class image on every alternating call
            assert names.count("k0s_pack_tiles") <= 2 and names.count("k1s_pack_x") <= 3, names
        _check_ll_only(us_a, m, gc, ut)
        _check_ll_only(us_b, m, gc, ut_c)
    ctx.set_k1_form("auto")
    for o in (us_b, us_a, tm, dm):
        o.close()
    ctx2.close()


@pytest.fixture(params=["pdf", "utt"])
def k1_fp32_form(request, opt):
    """The fp32-MFMA forms, whose contraction is pinned bit for bit to the k-ordered fmaf chain."""
    opt.k1(request.param)
    return request.param


@pytest.mark.parametrize("G,D", [(64, 40), (20, 13), (40, 80)])
def test_loglikes_long_utterances(ctx, G, D, k1_form):
    """Utterance lengths that give a wave 0..NF frame tiles, uneven remainders, partial last tiles
    and more than one chunk per utterance (features-only set, one shared pdf list)."""
    from kaldi_hmm_gmm_amd import DeviceModel, UtteranceSet

    P = 90 if G == 64 else 12       # 75 pdfs x 2 W tiles: a walk list longer than two 64-entry blocks
    m, gc, om, ut, cost = build(P, G, D, n_utt=2, seed=G + D, ragged=(G == 20))
    lens = np.array([1, 15, 16, 17, 63, 64, 65, 97, 160, 161, 255, 300, 320, 333, 384, 400, 480, 481, 700, 1111])
    frame_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rng = np.random.default_rng(G * 100 + D)
    feats = (rng.standard_normal((int(frame_off[-1]), D)) * 1.5 + 0.3).astype(np.float32)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    us = UtteranceSet(ctx, None, frame_off, feats)
    pl = np.sort(rng.choice(P, 75, replace=False)).astype(np.int32) if P == 90 else np.array([0, 3, 4, 7, 11], np.int32)
    us.set_pdf_list(pl)
    us.loglikes(dm)
    got = us.download_loglikes()
    for u in range(len(lens)):
        x = feats[frame_off[u]: frame_off[u + 1]]
        exact, bound = exact_loglikes(m, gc, x, pl)
        tol = LL_ATOL + LL_RTOL * bound
        assert got[u].shape == exact.shape
        assert np.isfinite(got[u]).all(), f"utt {u} (T={lens[u]}): non-finite log-likes"
        err = np.abs(got[u] - exact)
        assert (err <= tol).all(), f"utt {u} (T={lens[u]}): max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    us.close()


def test_loglikes_fma_order_bitwise_gemm(ctx, k1_fp32_form):
    """The MFMA contraction is bit-for-bit the k-ordered fmaf chain the oracle restates; only
    exp/log differ, so with G == 1 (log-sum-exp of one term == that term) results are bit-equal."""
    m, gc, om, ut, cost = build(30, 1, 40, n_utt=3, seed=5)
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


@pytest.mark.parametrize("beam,retry", [(200.0, 0.0), (6.0, 40.0), (1.0, 3.0)])
def test_align_bit_exact_on_same_scores(ctx, beam, retry):
    """K2 in isolation: feed the oracle's log-likes to the HIP Viterbi; alignment, words and
    status must equal the line-faithful FasterDecoder's exactly, `like` to float rounding."""
    m, gc, om, ut, cost = build(45, 4, 20, n_utt=40, seed=11, max_phones=7)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    poff, pdfs = us.pdf_lists()
    mats = [orc.loglikes_matrix(om, utt_feats(ut, u), pdfs[poff[u]: poff[u + 1]]) for u in range(us.n_utt)]
    us.upload_loglikes(mats)
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1)
    nfallback = 0
    for u in range(us.n_utt):
        g = oracle_graph(ut, u, cost)
        T = int(ut.frame_off[u + 1] - ut.frame_off[u])
        want = orc.align_utterance_ll(g, m.id2pdf, T, pdfs[poff[u]: poff[u + 1]], mats[u], acoustic_scale=0.1,
                                      beam=beam, retry_beam=retry)
        st = int(res["status"][u])
        nfallback += (st & 8) != 0
        assert (st & 3) == (want["status"] & 3), (u, st, want["status"])
        a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
        if want["status"] & 1:
            assert (a == 0).all()
        else:
            assert (a == want["ali"]).all(), f"utt {u} differs (status {st})"
            assert res["like"][u] == pytest.approx(want["like"], rel=1e-6, abs=1e-4)
    print(f"beam {beam}: {nfallback}/{us.n_utt} utterances used the faithful-decoder kernel")


def test_align_end_to_end(ctx):
    """K1 + K2 against oracle GMM decodable + FasterDecoder: identical alignments on this
    fixture (well separated models), `like` within the log-like tolerance."""
    m, gc, om, ut, cost = build(60, 8, 40, n_utt=16, seed=3, max_phones=6)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm)
    res = us.align(tm, acoustic_scale=0.1)
    for u in range(us.n_utt):
        g = oracle_graph(ut, u, cost)
        want = orc.align_utterance(g, om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1)
        a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
        assert int(res["status"][u]) & 1 == 0 and want["status"] == 0
        assert (a == want["ali"]).all()
        assert res["like"][u] == pytest.approx(want["like"], rel=2e-5)


@pytest.mark.parametrize("k3_form", ["wave", "wave_f32", "block", "valu"])
@pytest.mark.parametrize("P,G,D,ragged", [(30, 8, 39, True), (30, 64, 40, False), (30, 40, 13, True), (12, 128, 80, False), (10, 100, 77, True)])
def test_acc_stats_vs_oracle(ctx, P, G, D, ragged, k3_form, opt):
    """All three K3 accumulate kernels: the wave-local MFMA form (default for <= 64 Gaussians, D <= 40; its per-Gaussian
    log-likelihoods on the fp16 matrix cores in K1's f16x2s arithmetic -- "wave" -- or as the fp32 MFMA chain -- "wave_f32",
    option k3_phase_a = 1), the chunk-per-block MFMA form (option k3_form = 1; default for wider pdfs / features) and the
    VALU form (k3_form = 2; default above 128 Gaussians)."""
    from kaldi_hmm_gmm_amd import DeviceAccs

    if k3_form == "wave_f32":
        opt("k3_phase_a", 1)
    if k3_form == "block":
        opt("k3_form", 1)
    elif k3_form == "valu":
        opt("k3_form", 2)

    m, gc, om, ut, cost = build(P, G, D, n_utt=12, seed=7, ragged=ragged, max_phones=5)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.upload_ali(ut.ref_ali)
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs, weight=1.0)
    got = accs.download()
    oa = orc.OAccs(int(m.gauss_off[-1]), D, m.num_tids)
    tot = 0.0
    for u in range(us.n_utt):
        tot += orc.acc_stats_ali(om, m.id2pdf, utt_feats(ut, u), ut.ref_ali[ut.frame_off[u]: ut.frame_off[u + 1]], oa)
    assert (got["trans_acc"] == oa.trans_acc).all()
    assert got["trans_acc"].sum() == ut.frame_off[-1]        # scripts/test_gmm_acc_stats_ali.py:106
    assert got["total_frames"] == oa.total_frames
    assert got["total_log_like"] == pytest.approx(oa.total_log_like, rel=2e-6)
    np.testing.assert_allclose(got["occ"], oa.occ, rtol=2e-5, atol=1e-6)
    scale = np.abs(oa.mean_acc).max()
    np.testing.assert_allclose(got["mean_acc"], oa.mean_acc, rtol=2e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(got["var_acc"], oa.var_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.var_acc).max())


@pytest.mark.parametrize("D", [40, 39])
@pytest.mark.parametrize("big", [(130,), (70, 256), (300,)])
def test_acc_stats_form_is_chosen_per_pdf(ctx, big, D):
    """A model whose pdfs hold <= 64 Gaussians except for a few that Split (csrc/diag-gmm.cc:780-851) has grown: the wave form keeps
    every pdf it can take, only the grown ones go to the chunk-per-block MFMA form (<= 256 Gaussians) or the VALU form (beyond) --
    one launch per class, the same accumulator block.  Against the oracle (csrc/mle-diag-gmm.cc:123-158, csrc/diag-gmm.cc:368-392);
    the <= 64-Gaussian pdfs' statistics are bit-identical to those of the same pdfs in a model without the grown ones' frames."""
    from kaldi_hmm_gmm_amd import DeviceAccs
    P = 45
    counts = np.full(P, 64)
    counts[::7] = 33
    for k, g in enumerate(big):
        counts[5 + 11 * k] = g
    m, gc, om, ut, cost = build(P, 64, D, n_utt=40, seed=13, max_phones=6, gauss_counts=counts)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.upload_ali(ut.ref_ali)
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs, weight=1.0)
    got = accs.download()
    oa = orc.OAccs(int(m.gauss_off[-1]), D, m.num_tids)
    for u in range(us.n_utt):
        orc.acc_stats_ali(om, m.id2pdf, utt_feats(ut, u), ut.ref_ali[ut.frame_off[u]: ut.frame_off[u + 1]], oa)
    assert (got["trans_acc"] == oa.trans_acc).all() and got["total_frames"] == oa.total_frames
    assert got["total_log_like"] == pytest.approx(oa.total_log_like, rel=2e-6)
    np.testing.assert_allclose(got["occ"], oa.occ, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got["mean_acc"], oa.mean_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.mean_acc).max())
    np.testing.assert_allclose(got["var_acc"], oa.var_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.var_acc).max())
    per_pdf = np.bincount(m.id2pdf[ut.ref_ali], minlength=P)
    np.testing.assert_allclose(np.add.reduceat(got["occ"], m.gauss_off[:-1].astype(np.int64)), per_pdf, rtol=1e-5, atol=1e-4)
    assert per_pdf[[5 + 11 * k for k in range(len(big))]].min() > 0, "the grown pdfs hold frames"


@pytest.mark.parametrize("k3_form", ["wave", "wave_b64", "wave_f32", "block", "valu"])
@pytest.mark.parametrize("P,G,D", [(60, 64, 40), (300, 24, 39)])
def test_acc_stats_with_a_pdf_far_above_the_average(ctx, P, G, D, k3_form, opt):
    """Half of all phones are phone 0 (silence in real transcripts): its three pdfs hold ~15x / ~50x the frames of an average pdf, so
    k3_make_items cuts their buckets into more slices than the other pdfs' (one block per pdf would leave the chip waiting for three
    blocks).  Every K3 form against the oracle, and bit-reproducible run to run (slices are parked and added in slice order)."""
    from kaldi_hmm_gmm_amd import DeviceAccs

    if k3_form == "wave_b64":
        opt("k3_phase_b", 0)
    if k3_form == "wave_f32":
        opt("k3_phase_a", 1)
    if k3_form == "block":
        opt("k3_form", 1)
    elif k3_form == "valu":
        opt("k3_form", 2)
    m, gc, om, ut, cost = build(P, G, D, n_utt=60, seed=11, min_phones=20, max_phones=40, transcripts="skew")
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.upload_ali(ut.ref_ali)
    per_pdf = np.bincount(m.id2pdf[ut.ref_ali], minlength=P)
    assert per_pdf[:3].min() > 8 * per_pdf.mean()
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs, weight=1.0)
    got = accs.download()
    oa = orc.OAccs(int(m.gauss_off[-1]), D, m.num_tids)
    for u in range(us.n_utt):
        orc.acc_stats_ali(om, m.id2pdf, utt_feats(ut, u), ut.ref_ali[ut.frame_off[u]: ut.frame_off[u + 1]], oa)
    assert (got["trans_acc"] == oa.trans_acc).all() and got["total_frames"] == oa.total_frames
    assert got["total_log_like"] == pytest.approx(oa.total_log_like, rel=2e-6)
    np.testing.assert_allclose(got["occ"], oa.occ, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got["mean_acc"], oa.mean_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.mean_acc).max())
    np.testing.assert_allclose(got["var_acc"], oa.var_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.var_acc).max())
    per_gauss = np.add.reduceat(got["occ"], m.gauss_off[:-1].astype(np.int64))
    np.testing.assert_allclose(per_gauss, per_pdf, rtol=1e-5, atol=1e-4)
    if k3_form.startswith("wave"):              # the block / VALU forms end in fp64 atomics: sums in any order
        accs2 = DeviceAccs(ctx, dm, tm)
        us.acc_stats(dm, tm, accs2, weight=1.0)
        again = accs2.download()
        for k in ("occ", "mean_acc", "var_acc", "trans_acc"):
            assert np.array_equal(got[k], again[k]), k


@pytest.mark.parametrize("phase_b", [0, 2])
@pytest.mark.parametrize("overlap", [False, True])
def test_acc_stats_fp16_phase_a_against_fp64_posteriors_and_sharding(ctx, opt, overlap, phase_b):
    """K3's phase A on the fp16 matrix cores (f16x2s split, scale exponents from the model alone): its statistics lie as close to
    an fp64 evaluation of the posteriors as the fp32 chain's do, and -- the exponents not depending on the utterances -- two
    shards of a set add up to the statistics of the whole set: to fp64 rounding with phase B on the fp64 pipe (k3_phase_b = 0: exact
    products), to ~1e-7 of a cell's magnitude with phase B on the fp16 matrix cores (the default, k3_phase_b = 2: 32-frame fp32
    partial sums, and which frames share a group depends on the shard); occ is exact either way."""
    opt("k3_phase_b", phase_b)
    from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet

    m, gc, om, ut, cost = build(20, 64, 40, n_utt=24, seed=21, max_phones=6)
    if overlap:                                   # posteriors spread over many components
        means = (0.12 * m.means).astype(np.float32)
        miv = (means * m.inv_vars).astype(np.float32)
        gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, miv)
    else:
        miv = m.means_invvars
    dm = DeviceModel(ctx, m.gauss_off, gc, miv, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)

    def run(sel, phase_a):
        opt("k3_phase_a", phase_a)
        fo = np.concatenate([[0], np.cumsum(np.diff(ut.frame_off)[sel])]).astype(np.int64)
        rows = np.concatenate([np.arange(ut.frame_off[u], ut.frame_off[u + 1]) for u in sel])
        us = UtteranceSet(ctx, tm, fo, np.ascontiguousarray(ut.feats[rows]))
        us.upload_ali(np.ascontiguousarray(ut.ref_ali[rows]))
        accs = DeviceAccs(ctx, dm, tm)
        us.acc_stats(dm, tm, accs, weight=1.0)
        return accs.download()

    every = np.arange(len(ut.frame_off) - 1)
    f16 = run(every, 0)
    f32 = run(every, 1)
    # fp64 posteriors and sums on the host
    X = ut.feats.astype(np.float64)
    pdf = m.id2pdf[ut.ref_ali]
    occ = np.zeros(int(m.gauss_off[-1])); macc = np.zeros((occ.size, 40)); tot = 0.0
    for p in np.unique(pdf):
        g0, g1 = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        x = X[pdf == p]
        ll = gc[g0:g1].astype(np.float64)[None, :] + x @ miv[g0:g1].astype(np.float64).T - 0.5 * (x * x) @ m.inv_vars[g0:g1].astype(np.float64).T
        mx = ll.max(1, keepdims=True)
        e = np.exp(ll - mx); z = e.sum(1, keepdims=True)
        post = e / z
        tot += float((np.log(z) + mx).sum())
        occ[g0:g1] = post.sum(0); macc[g0:g1] = post.T @ x
    big = occ > 1e-3
    err16 = np.abs(f16["occ"] - occ)[big] / occ[big]
    err32 = np.abs(f32["occ"] - occ)[big] / occ[big]
    assert err16.max() <= max(2.0 * err32.max(), 2e-6), (err16.max(), err32.max())
    assert f16["total_log_like"] == pytest.approx(tot, rel=2e-6)
    e16 = np.abs(f16["mean_acc"] - macc).max(); e32 = np.abs(f32["mean_acc"] - macc).max()
    assert e16 <= max(2.0 * e32, 2e-6 * np.abs(macc).max()), (e16, e32)
    # sharding: halves of the set, fp16 phase A on both
    a = run(every[::2], 0); b = run(every[1::2], 0)
    tol = 1e-11 if phase_b == 0 else 1e-6
    for k in ("occ", "mean_acc", "var_acc"):
        np.testing.assert_allclose(a[k] + b[k], f16[k], rtol=1e-11 if k == "occ" else tol, atol=(1e-11 if k == "occ" else tol) * np.abs(f16[k]).max())
    assert a["total_log_like"] + b["total_log_like"] == pytest.approx(f16["total_log_like"], rel=1e-12)


@pytest.mark.parametrize("k3_form", ["wave", "wave_f32", "block", "valu"])
@pytest.mark.parametrize("P,G,D", [(30, 64, 40), (30, 24, 23), (12, 128, 80)])
def test_acc_stats_overlapping_gaussians_vs_oracle(ctx, P, G, D, k3_form, opt):
    """The synthetic model's Gaussians are ~27 sigma apart (posteriors are one-hot, any softmax would do); here the
    means are pulled together so every frame spreads its posterior over many components and the softmax, the
    per-frame normalisation and the gamma-weighted sums are all exercised against the oracle."""
    from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet

    if k3_form == "wave_f32":
        opt("k3_phase_a", 1)
    if k3_form == "block":
        opt("k3_form", 1)
    elif k3_form == "valu":
        opt("k3_form", 2)
    m, _, _, ut, cost = build(P, G, D, n_utt=12, seed=13, max_phones=5)
    means = (0.12 * m.means).astype(np.float32)
    miv = (means * m.inv_vars).astype(np.float32)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, miv)
    om = orc.OModel(m.gauss_off, gc, miv, m.inv_vars)
    feats = (0.12 * ut.feats + np.random.default_rng(2).standard_normal(ut.feats.shape)).astype(np.float32)
    dm = DeviceModel(ctx, m.gauss_off, gc, miv, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    us = UtteranceSet(ctx, tm, ut.frame_off, feats)
    us.upload_ali(ut.ref_ali)
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs, weight=0.75)
    got = accs.download()
    oa = orc.OAccs(int(m.gauss_off[-1]), D, m.num_tids)
    for u in range(us.n_utt):
        sl = slice(ut.frame_off[u], ut.frame_off[u + 1])
        orc.acc_stats_ali(om, m.id2pdf, feats[sl], ut.ref_ali[sl], oa, weight=0.75)
    # the posteriors really are diffuse: the largest component takes well under all of a frame's mass on average
    per_pdf_max = np.maximum.reduceat(oa.occ, m.gauss_off[:-1].astype(np.int64))
    per_pdf_sum = np.add.reduceat(oa.occ, m.gauss_off[:-1].astype(np.int64))
    assert (per_pdf_max[per_pdf_sum > 0] / per_pdf_sum[per_pdf_sum > 0]).mean() < 0.5
    assert got["total_frames"] == pytest.approx(oa.total_frames, rel=1e-12)
    assert got["total_log_like"] == pytest.approx(oa.total_log_like, rel=2e-6)
    # Stated tolerance for diffuse posteriors: a posterior is exp(ll_g - ll) of two fp32 log-likelihoods, each within
    # 1e-5 + 1e-6*B (B ~ 60 here) of the exact value in EITHER implementation (different summation orders of the
    # D-term dot products), so gamma -- and sums of gammas -- agree to ~1e-4 relative, not to the 2e-5 of the
    # one-hot case above.
    np.testing.assert_allclose(got["occ"], oa.occ, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(got["mean_acc"], oa.mean_acc, rtol=2e-4, atol=2e-5 * np.abs(oa.mean_acc).max())
    np.testing.assert_allclose(got["var_acc"], oa.var_acc, rtol=2e-4, atol=2e-5 * np.abs(oa.var_acc).max())
    # and against an fp64 evaluation of the same posteriors (what both approximate) at the same tolerance
    x = feats.astype(np.float64)
    pdf = m.id2pdf[ut.ref_ali]
    occ64 = np.zeros_like(oa.occ)
    for p in np.unique(pdf):
        a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        xs = x[pdf == p]
        ll = gc[a:b].astype(np.float64)[None, :] + xs @ miv[a:b].astype(np.float64).T - 0.5 * (xs * xs) @ m.inv_vars[a:b].astype(np.float64).T
        g = np.exp(ll - ll.max(1, keepdims=True))
        occ64[a:b] = 0.75 * (g / g.sum(1, keepdims=True)).sum(0)
    np.testing.assert_allclose(got["occ"], occ64, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("P,G,D,n_utt", [(12, 40, 23, 60), (300, 8, 13, 400), (3, 4, 40, 5)])
def test_counting_sort_bucketing_equals_radix_sort_bit_for_bit(ctx, opt, P, G, D, n_utt):
    """The library's own stable counting sort of the frames by pdf (k3_cs_hist / _scan / _starts / _place, k3_bucket = 2) puts
    every frame where rocPRIM's stable radix sort of (pdf, frame) pairs does (the default): the statistics of the wave form,
    which sums a pdf's frames in bucket order, are bit-identical; frames without a valid alignment (transition-id 0) stay out."""
    from kaldi_hmm_gmm_amd import DeviceAccs

    m, gc, om, ut, cost = build(P, G, D, n_utt=n_utt, seed=P + 1, ragged=True, max_phones=6)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    ali = ut.ref_ali.copy()
    ali[::17] = 0                                   # unaligned frames scattered through the set
    us.upload_ali(ali)
    out = []
    for mode in (0, 2, 0):
        opt("k3_bucket", mode)
        accs = DeviceAccs(ctx, dm, tm)
        us.acc_stats(dm, tm, accs, weight=0.7)
        st = accs.download()
        out.append(np.concatenate([st["occ"], st["mean_acc"].ravel(), st["var_acc"].ravel(), st["trans_acc"],
                                   [st["total_frames"], st["total_log_like"]]]))
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
    assert out[0][-2] == pytest.approx(float(np.float32(0.7)) * (ali != 0).sum(), rel=1e-12)


def test_acc_stats_reproducible_bit_for_bit(ctx, opt):
    """Wave-form K3 (the default for <= 64 Gaussians, D <= 40): stable bucket sort, per-pdf tile order, waves and pdf
    slices folded in a fixed order, one atomic per cell -- repeated passes give identical bits, with one block per pdf
    and with a pdf cut into several blocks (small models), scalars included."""
    from kaldi_hmm_gmm_amd import DeviceAccs

    m, gc, om, ut, cost = build(12, 40, 23, n_utt=60, seed=21, ragged=True, max_phones=8)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.upload_ali(ut.ref_ali)
    for ny in ("1", "7"):
        opt("k3_ny", int(ny))
        runs = []
        for _ in range(3):
            accs = DeviceAccs(ctx, dm, tm)
            us.acc_stats(dm, tm, accs, weight=0.3)
            buf = np.zeros(accs.size, np.float64)
            st = accs.download()
            runs.append(np.concatenate([st["occ"], st["mean_acc"].ravel(), st["var_acc"].ravel(), st["trans_acc"],
                                        [st["total_frames"], st["total_log_like"]]]))
        assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2]), ny
        assert runs[0][-2] == pytest.approx(0.3 * ut.frame_off[-1], rel=1e-6)


def test_align_very_short_utterances(ctx):
    """T = 3 .. ~12 frames (one to three phones, almost no self-loops): the packed back-pointer groups of eight layers
    are partial, the score blocks are mostly padding, and utterances one frame too short for their graph must fail
    exactly like the reference."""
    m, gc, om, ut, cost = build(30, 4, 13, n_utt=40, seed=77, min_phones=1, max_phones=3)
    ut = synth.make_utts(m, 40, seed=78, min_phones=1, max_phones=3, leave_prob=0.97)
    T = np.diff(ut.frame_off)
    assert T.min() <= 4 and T.max() <= 16
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=200.0, acoustic_scale=0.1)
    for u in range(us.n_utt):
        want = orc.align_utterance(oracle_graph(ut, u, cost), om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1)
        assert (int(res["status"][u]) & 1) == (want["status"] & 1)
        if not want["status"] & 1:
            assert (res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]] == want["ali"]).all()
    # drop the last frame of every utterance: the ones that had exactly one frame per state can no longer reach the end
    fo = ut.frame_off - np.arange(len(ut.frame_off))
    keep = np.ones(ut.frame_off[-1], bool)
    keep[ut.frame_off[1:] - 1] = False
    import dataclasses
    ut2 = dataclasses.replace(ut, frame_off=fo.astype(np.int64), feats=ut.feats[keep])
    dm2, tm2, us2 = _device(ctx, m, gc, ut2, cost)
    us2.loglikes(dm2, reachable_only=True)
    res2 = us2.align(tm2, beam=200.0, acoustic_scale=0.1)
    nerr = 0
    for u in range(us2.n_utt):
        f = utt_feats(ut2, u)
        want = orc.align_utterance(oracle_graph(ut2, u, cost), om, m.id2pdf, f, acoustic_scale=0.1)
        assert (int(res2["status"][u]) & 1) == (want["status"] & 1), u
        nerr += want["status"] & 1
        if not want["status"] & 1:
            assert (res2["ali"][ut2.frame_off[u]: ut2.frame_off[u + 1]] == want["ali"]).all()
    assert nerr > 0


def test_align_long_utterances_many_waves(ctx):
    """600-900 states (10-15 waves per utterance), thousands of frames: wave trimming per utterance, the 32-layer
    certificate folds and the back-pointer groups across many waves."""
    m, gc, om, ut, cost = build(90, 3, 13, n_utt=4, seed=31, min_phones=200, max_phones=300)
    S = np.diff(ut.graphs["state_off"])
    assert S.max() > 640 and np.diff(ut.frame_off).max() > 2000
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm, reachable_only=True)
    for beam, retry in ((200.0, 0.0), (8.0, 40.0)):
        res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1)
        for u in range(us.n_utt):
            want = orc.align_utterance(oracle_graph(ut, u, cost), om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1, beam=beam,
                                       retry_beam=retry)
            assert want["status"] & 1 == 0 and int(res["status"][u]) & 1 == 0
            assert (res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]] == want["ali"]).all()
            assert res["like"][u] == pytest.approx(want["like"], rel=2e-5)


def test_align_graphs_beyond_1024_states(ctx):
    """More than 1024 states: two states per thread on K2's register-resident path; a 2700-state training graph, whose
    decoder tables no longer fit the LDS (round 1 refused it), runs from the HBM scratch slice -- same alignment as the oracle."""

    m, gc, om, ut, cost = build(90, 2, 13, n_utt=2, seed=400, min_phones=400, max_phones=420)
    S = int(np.diff(ut.graphs["state_off"]).max())
    assert 1024 < S <= 2048
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=200.0, acoustic_scale=0.1)
    for u in range(us.n_utt):
        want = orc.align_utterance(oracle_graph(ut, u, cost), om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1)
        assert want["status"] & 1 == 0 and int(res["status"][u]) & 1 == 0
        assert (res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]] == want["ali"]).all()
        assert res["like"][u] == pytest.approx(want["like"], rel=2e-5)
    # four states per thread, tables still in LDS
    m, gc, om, ut, cost = build(90, 2, 13, n_utt=2, seed=700, min_phones=700, max_phones=730)
    S = int(np.diff(ut.graphs["state_off"]).max())
    assert 2048 < S <= 2200
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=200.0, acoustic_scale=0.1)
    for u in range(us.n_utt):
        want = orc.align_utterance(oracle_graph(ut, u, cost), om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1)
        assert want["status"] & 1 == 0 and int(res["status"][u]) & 1 == 0 and int(res["status"][u]) & 4
        assert (res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]] == want["ali"]).all()
        assert res["like"][u] == pytest.approx(want["like"], rel=2e-5)
    m, gc, om, ut, cost = build(90, 2, 13, n_utt=1, seed=900, min_phones=900, max_phones=910)
    assert int(np.diff(ut.graphs["state_off"]).max()) > 2500
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=200.0, acoustic_scale=0.1)
    want = orc.align_utterance(oracle_graph(ut, 0, cost), om, m.id2pdf, utt_feats(ut, 0), acoustic_scale=0.1)
    assert want["status"] & 1 == 0 and int(res["status"][0]) & 1 == 0
    assert (res["ali"] == want["ali"]).all() and res["like"][0] == pytest.approx(want["like"], rel=2e-5)


def _first_frames(g, u, id2pdf, pdfs):
    """Fewest emitting arcs before an arc with each listed pdf can be taken (0-1 BFS from the start state)."""
    from collections import deque
    s0, s1 = int(g["state_off"][u]), int(g["state_off"][u + 1])
    S = s1 - s0
    ao = g["arc_off"]
    dmin = [None] * S
    st = int(g["start"][u])
    dmin[st] = 0
    q = deque([st])
    while q:
        s = q.popleft()
        for a in range(int(ao[s0 + s]), int(ao[s0 + s + 1])):
            d, w = int(g["nextstate"][a]), 1 if g["ilabel"][a] >= 1 else 0
            if dmin[d] is None or dmin[s] + w < dmin[d]:
                dmin[d] = dmin[s] + w
                (q.append if w else q.appendleft)(d)
    first = {int(p): 10**9 for p in pdfs}
    for s in range(S):
        if dmin[s] is None:
            continue
        for a in range(int(ao[s0 + s]), int(ao[s0 + s + 1])):
            if g["ilabel"][a] >= 1:
                p = int(id2pdf[g["ilabel"][a]])
                first[p] = min(first[p], dmin[s])
    return first


def test_reachable_only_loglikes_skip_only_unreadable_cells(ctx, k1_form):
    """khg_loglikes_reachable: every cell a decoder token can read is bit-identical to the full matrix, whole
    16-frame tiles before a pdf's first readable frame are left untouched, and the alignment is unchanged."""
    m, gc, om, ut, cost = build(150, 64, 40, n_utt=10, seed=77, min_phones=12, max_phones=40)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm)
    full = us.download_loglikes()
    res_full = us.align(tm, beam=200.0, retry_beam=0.0, acoustic_scale=0.1)
    poff, pdfs = us.pdf_lists()
    POISON = np.float32(12345.0)
    us.upload_loglikes([np.full_like(f, POISON) for f in full])
    us.loglikes(dm, reachable_only=True)
    part = us.download_loglikes()
    skipped = 0
    for u in range(us.n_utt):
        pl = pdfs[poff[u]: poff[u + 1]]
        first = _first_frames(ut.graphs, u, m.id2pdf, pl)
        for j, p in enumerate(pl):
            t0 = min(first[int(p)], full[u].shape[1])          # (in front of it: whole tiles skipped, or a tile that starts at this frame)
            assert np.array_equal(part[u][j, t0:], full[u][j, t0:]), (u, j)
            untouched = part[u][j, :t0] == POISON
            assert (untouched | (part[u][j, :t0] == full[u][j, :t0])).all()     # skipped, or computed exactly
            skipped += int(untouched.sum())
    total = sum(f.size for f in full)
    assert skipped > 0.03 * total, (skipped, total)                              # the early triangle really is skipped
    res = us.align(tm, beam=200.0, retry_beam=0.0, acoustic_scale=0.1)
    assert np.array_equal(res["ali"], res_full["ali"]) and np.array_equal(res["status"], res_full["status"])
    np.testing.assert_array_equal(res["like"], res_full["like"])
    print("skipped fraction", skipped / total)


def test_reference_stored_logsumexp_softmax_vectors_through_k1_and_k3(ctx):
    """The only literal stored answers the reference holds for LogSumExp / Softmax (csrc/eigen-test.cc:460-475, :641-655),
    through the product's kernels: a 1-dim pdf with zero means_invvars / inv_vars has component log-likelihoods ==
    gconsts, so K1's fused log-sum-exp must give 1.8119 / 2.1343 and K3's posteriors the stored Softmax vector, at the
    reference's own tolerance (1e-4) -- and agree with the oracle to float rounding."""
    from kaldi_hmm_gmm_amd import _gpu
    from test_oracle_pins import LSE_V10, LSE_V5, SOFTMAX_EXPECTED, SOFTMAX_V

    _gpu.set_default_context(ctx)
    x = np.array([[0.7], [-1.3], [0.0]], np.float32)
    for v, want in ((LSE_V5, 1.8119), (LSE_V10, 2.1343)):
        g = np.asarray(v, np.float32)
        z = np.zeros((len(v), 1), np.float32)
        ll = _gpu.loglikes(np.array([0, len(v)], np.int32), g, z, z, x, [0])
        assert ll.shape == (1, 3) and (np.abs(ll - want) < 1e-4).all()
        assert (np.abs(ll - orc.logsumexp(v)) <= 2e-6).all()
    g = np.asarray(SOFTMAX_V, np.float32)
    z = np.zeros((5, 1), np.float32)
    st = _gpu.acc_stats(np.array([0, 5], np.int32), g, z, z, x[:1], [0])
    assert np.abs(st["occ"] - np.asarray(SOFTMAX_EXPECTED)).max() < 1e-4
    post, lse = orc.softmax(SOFTMAX_V)
    assert np.abs(st["occ"] - post).max() <= 2e-7 and abs(st["total_log_like"] - lse) <= 2e-6


def test_set_pdf_list_twice_on_one_set(ctx, k1_form):
    """khg_utts_set_pdf_list called again on the same set -- first a short list, then a LONGER one with other pdfs -- must score the
    new list: every per-set cache derived from the old list (tile walk lists, the default form's unit table, the id range check) is
    dropped.  A pdf id outside the model is an error the second time too."""
    from kaldi_hmm_gmm_amd import DeviceModel, KhgError, UtteranceSet

    P, G, D = 40, 64, 40
    m, gc, om, ut, cost = build(P, G, D, n_utt=2, seed=77)
    lens = np.array([33, 200, 64, 7])
    frame_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rng = np.random.default_rng(5)
    feats = (rng.standard_normal((int(frame_off[-1]), D)) * 1.5 + 0.3).astype(np.float32)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    us = UtteranceSet(ctx, None, frame_off, feats)
    for pl in (np.array([3, 9, 17], np.int32), np.array([0, 1, 2, 5, 8, 13, 21, 22, 34, 39], np.int32), np.array([38], np.int32)):
        us.set_pdf_list(pl)
        us.loglikes(dm)
        got = us.download_loglikes()
        for u in range(len(lens)):
            x = feats[frame_off[u]: frame_off[u + 1]]
            exact, bound = exact_loglikes(m, gc, x, pl)
            assert got[u].shape == exact.shape
            assert (np.abs(got[u] - exact) <= LL_ATOL + LL_RTOL * bound).all(), f"list {pl.tolist()} utt {u}"
    us.set_pdf_list(np.array([1, P], np.int32))
    with pytest.raises(KhgError):
        us.loglikes(dm)
    us.close()


def test_features_changed_repacks_the_planes(ctx, opt):
    """Borrowed device features rewritten in place + khg_utts_features_changed: the default K1 (fp16 planes packed once per set)
    scores the NEW features."""
    import torch

    from kaldi_hmm_gmm_amd import DeviceModel, UtteranceSet

    opt.k1("f16x2s")
    P, G, D = 12, 64, 40
    m, gc, om, ut, cost = build(P, G, D, n_utt=2, seed=78)
    frame_off = np.array([0, 100, 164], np.int64)
    rng = np.random.default_rng(6)
    f1 = (rng.standard_normal((164, D)) * 1.5).astype(np.float32)
    f2 = (rng.standard_normal((164, D)) * 2.5 + 1.0).astype(np.float32)
    buf = torch.from_numpy(f1).cuda()
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    us = UtteranceSet(ctx, None, frame_off, (buf.data_ptr(), buf), dim=D)
    pl = np.arange(P, dtype=np.int32)
    us.set_pdf_list(pl)
    for f in (f1, f2):
        buf.copy_(torch.from_numpy(f))
        torch.cuda.synchronize()
        us.features_changed()
        us.loglikes(dm)
        got = us.download_loglikes()
        for u in range(2):
            x = f[frame_off[u]: frame_off[u + 1]]
            exact, bound = exact_loglikes(m, gc, x, pl)
            assert (np.abs(got[u] - exact) <= LL_ATOL + LL_RTOL * bound).all()
    us.close()


def _last_frames(graphs, u, id2pdf, pdf_list, T):
    """last frame at which an arc carrying each pdf can still lead to a final state by frame T (host restatement of pdf_last)."""
    g = graphs
    s0, s1 = int(g["state_off"][u]), int(g["state_off"][u + 1])
    S = s1 - s0
    ao = g["arc_off"]
    dfin = np.full(S, 10**9, np.int64)
    fin = np.isfinite(g["final"][s0:s1])
    dfin[fin] = 0
    changed = True
    while changed:                      # Bellman-Ford on the reversed graph (tiny graphs): emitting arcs weigh 1, epsilons 0
        changed = False
        for s in range(S):
            for a in range(int(ao[s0 + s]), int(ao[s0 + s + 1])):
                d, w = int(g["nextstate"][a]), int(g["ilabel"][a] >= 1)
                if dfin[d] + w < dfin[s]:
                    dfin[s] = dfin[d] + w; changed = True
    last = {int(p): -1 for p in pdf_list}
    for s in range(S):
        for a in range(int(ao[s0 + s]), int(ao[s0 + s + 1])):
            if g["ilabel"][a] >= 1 and dfin[int(g["nextstate"][a])] < 10**9:
                p = int(id2pdf[g["ilabel"][a]])
                last[p] = max(last[p], T - 1 - int(dfin[int(g["nextstate"][a])]))
    return last


@pytest.mark.parametrize("G,D", [(64, 40), (150, 40), (100, 72)])
@pytest.mark.parametrize("beam,retry,max_active", [(200.0, 0.0, 2**31 - 1), (2.0, 30.0, 2**31 - 1), (200.0, 0.0, 100000), (64.0, 0.0, 2**31 - 1)])
def test_band_loglikes_fill_only_dead_cells_and_align_identically(ctx, opt, beam, retry, max_active, G, D):
    """khg_loglikes_band (default K1 form): between a pdf's first and last needed 32-frame tile the scores are bit-identical to the
    full matrix; whole tiles past the last needed one hold an UPPER BOUND of the pdf's log-likelihood (>= every true value of that
    pdf); and khg_align gives the alignment of the full matrix -- through the exact DP + certificate, or (max_active
    set: the DP certifies nothing, every utterance goes this way) through the repair launch + order-faithful decoder -- which is
    the oracle's."""
    # (G, D) = (150, 40): five W tiles = three passes of two per pdf, later passes combine with the stored value at the band's own,
    # possibly shifted, cells; (100, 72): KS = 10, one W tile per pass, four passes
    opt.k1("f16x2s")
    if (G, D) != (64, 40) and (retry != 0.0 or beam == 64.0):
        pytest.skip("the multi-pass shapes run the wide-beam and the all-repaired cases")
    m, gc, om, ut, cost = build(150, G, D, n_utt=12, seed=78, min_phones=12, max_phones=40)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm)
    full = us.download_loglikes()
    res_full = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1, max_active=max_active)
    poff, pdfs = us.pdf_lists()
    last_dev = us.pdf_last_frames()
    first_dev = us.pdf_first_frames()
    POISON = np.float32(12345.0)
    us.upload_loglikes([np.full_like(f, POISON) for f in full])
    us.loglikes(dm, band=True)
    part = us.download_loglikes()
    filled = total = shifted = 0
    for u in range(us.n_utt):
        pl = pdfs[poff[u]: poff[u + 1]]
        T = int(ut.frame_off[u + 1] - ut.frame_off[u])
        first = _first_frames(ut.graphs, u, m.id2pdf, pl)
        last = _last_frames(ut.graphs, u, m.id2pdf, pl, T)
        assert [last[int(p)] for p in pl] == last_dev[poff[u]: poff[u + 1]].tolist()
        for j, p in enumerate(pl):
            fp, lp = min(first[int(p)], 10**6), max(last[int(p)], 0)
            fd = int(first_dev[poff[u] + j])                 # the library's own first readable frame (never later than the graph's)
            assert fd <= fp
            fp = fd
            t0 = 32 * (fp // 32)
            t1 = min(32 * (lp // 32 + 1), full[u].shape[1])
            if last[int(p)] >= 0 and lp % 32 < fp % 32 and lp // 32 > fp // 32 and (part[u][j, t0:fp] == POISON).all():
                # shifted tiles: the band ends earlier inside its tile than it starts -- its tiles start at its first frame, one fewer
                # of them; the frames in front stay untouched.  (A band that crosses the 480-frame chunks of a long utterance keeps
                # the aligned tiles.)
                assert T <= 480
                t0, t1 = fp, fp + 32 * (lp // 32 - fp // 32)
                shifted += 1
            assert fp >= t0 and (last[int(p)] < 0 or lp < t1)
            assert np.array_equal(part[u][j, t0:t1], full[u][j, t0:t1]), (u, j)
            tail = part[u][j, max(t1, t0):]
            if tail.size:
                assert (tail == tail[0]).all() and tail[0] != POISON, "tiles past the band hold one fill value"
                assert tail[0] >= full[u][j].max() and tail[0] >= full[u][j, max(t1, t0):].max(), "... an upper bound of the pdf's scores"
                filled += tail.size
            total += full[u].shape[1]
    assert filled > 0.05 * total, (filled, total)
    assert shifted > 40, "hardly a band took the shifted tiles"
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1, max_active=max_active)
    assert np.array_equal(res["ali"], res_full["ali"]) and np.array_equal(res["status"] & 3, res_full["status"] & 3)
    np.testing.assert_array_equal(res["like"], res_full["like"])
    nfb = int(((res["status"] & 8) != 0).sum())
    if max_active < 2**31 - 1:
        assert nfb == us.n_utt, "max_active was meant to send every utterance through the repair launch"
        # after the repair the flagged utterances hold real scores from their first needed tile on
        part2 = us.download_loglikes()
        for u in np.nonzero(res["status"] & 8)[0]:
            pl = pdfs[poff[u]: poff[u + 1]]
            first = _first_frames(ut.graphs, int(u), m.id2pdf, pl)
            for j, p in enumerate(pl):
                t0 = 32 * (min(first[int(p)], 10**6) // 32)
                assert np.array_equal(part2[u][j, t0:], full[u][j, t0:]), (u, j)
    for u in range(us.n_utt):
        want = orc.align_utterance(oracle_graph(ut, u, cost), om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1, beam=beam, retry_beam=retry,
                                   max_active=max_active)
        a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
        assert (int(res["status"][u]) & 3) == (want["status"] & 3)
        if not want["status"] & 1:
            assert (a == want["ali"]).all(), u
    print(f"beam {beam}: band fill fraction {filled / total:.3f}, {nfb}/{us.n_utt} utterances repaired + decoded order-faithfully")


@pytest.mark.parametrize("G,weight", [(64, 1.0), (50, 0.7), (40, 3.0e-4), (64, -2.5)])
def test_acc_stats_fp16_phase_b_vs_oracle_and_fp64_form(ctx, opt, G, weight):
    """K3 with phase B on the fp16 matrix cores too (option k3_phase_b = 2; k3_accumulate_wave16): gamma and [x | x^2] split into two
    fp16 pieces each, three partial products, 32-frame fp32 partial sums added in fp64.  Against the oracle at the tolerance of every
    other form (2e-5), against the fp64 form much closer (a product carries 22 + 22 bits, 32 of them are summed in fp32: ~1e-7 of the
    cell's magnitude), occ / transition counts / total log-like computed as in the fp64 form; one block per pdf and slices; weights of
    any magnitude and sign (the gamma scale 2^SG follows the weight)."""
    from kaldi_hmm_gmm_amd import DeviceAccs

    m, gc, om, ut, cost = build(20, G, 40, n_utt=80, seed=51 + G, ragged=(G == 50), max_phones=8)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.upload_ali(ut.ref_ali)
    oa = orc.OAccs(int(m.gauss_off[-1]), m.dim, m.num_tids)
    for u in range(us.n_utt):
        orc.acc_stats_ali(om, m.id2pdf, utt_feats(ut, u), ut.ref_ali[ut.frame_off[u]: ut.frame_off[u + 1]], oa, weight=weight)
    for ny in (1, 5):
        opt("k3_ny", ny)
        st = {}
        for pb in (0, 2):
            opt("k3_phase_b", pb)
            runs = []
            for _ in range(2):
                accs = DeviceAccs(ctx, dm, tm)
                us.acc_stats(dm, tm, accs, weight=weight)
                runs.append(accs.download())
            for k in ("occ", "mean_acc", "var_acc", "trans_acc"):
                assert np.array_equal(runs[0][k], runs[1][k]), (pb, k)            # run-to-run reproducible
            st[pb] = runs[0]
        a, b = st[0], st[2]
        assert np.array_equal(a["occ"], b["occ"]) and np.array_equal(a["trans_acc"], b["trans_acc"]) and a["total_frames"] == b["total_frames"]
        assert b["total_log_like"] == pytest.approx(a["total_log_like"], rel=1e-12)
        mm, vm = np.abs(a["mean_acc"]).max(), np.abs(a["var_acc"]).max()
        np.testing.assert_allclose(b["mean_acc"], a["mean_acc"], rtol=2e-6, atol=2e-7 * mm)
        np.testing.assert_allclose(b["var_acc"], a["var_acc"], rtol=2e-6, atol=2e-7 * vm)
        np.testing.assert_allclose(b["occ"], oa.occ, rtol=2e-5, atol=1e-6 * abs(weight))
        np.testing.assert_allclose(b["mean_acc"], oa.mean_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.mean_acc).max())
        np.testing.assert_allclose(b["var_acc"], oa.var_acc, rtol=2e-5, atol=2e-6 * np.abs(oa.var_acc).max())


def test_band_fill_bounds_k1_values_for_unnormalised_features(ctx, opt):
    """The band form fills dead cells with log sum_g exp(gconst_g + 0.5 sum_d mi^2 / iv) + a margin.  With un-normalised features
    (means of ~20 standard deviations: gconst and the quadratic terms are ~1e4 each and cancel) an fp32 evaluation of that sum, or
    a margin that ignores the magnitude of the cancelling terms, falls BELOW values K1 itself computes near a component's mean
    (round-4 advisor finding): the bound is summed in fp64 and its margin scales with the terms (k0_model_stats)."""
    from kaldi_hmm_gmm_amd import synth
    opt.k1("f16x2s")
    m = synth.make_model(90, 40, 40, seed=11)
    m.means[:] = (m.means * np.float32(2.0) + np.float32(15.0)).astype(np.float32)      # (far larger offsets leave the f16x2s domain: no band at all)
    m.means_invvars[:] = (m.means * m.inv_vars).astype(np.float32)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    ut = synth.make_utts(m, 10, seed=4, min_phones=10, max_phones=30)
    cost = np.zeros(m.num_tids + 1, np.float32)
    dm, tm, us = _device(ctx, m, gc, ut, cost)
    us.loglikes(dm)
    full = us.download_loglikes()
    us.loglikes(dm, band=True)
    part = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    last_dev = us.pdf_last_frames()
    checked = 0
    for u in range(us.n_utt):
        T = int(ut.frame_off[u + 1] - ut.frame_off[u])
        for j in range(int(poff[u + 1] - poff[u])):
            lp = int(last_dev[poff[u] + j])
            t1 = min(32 * (max(lp, 0) // 32 + 1), full[u].shape[1])
            tail = part[u][j, t1:T]
            differs = tail != full[u][j, t1:T]
            if differs.any():                                  # a filled tile: one value, no smaller than anything K1 computes for this pdf
                fill = tail[differs][0]
                assert (tail[differs] == fill).all()
                assert fill >= full[u][j, :T].max(), (u, j, float(fill), float(full[u][j, :T].max()))
                checked += 1
    assert checked > 50
    # and the frames a pdf's own component emitted score close to that bound: the margin is not vacuous headroom
    res_b = us.align(tm, beam=200.0, acoustic_scale=0.1)
    us.loglikes(dm)
    res_f = us.align(tm, beam=200.0, acoustic_scale=0.1)
    np.testing.assert_array_equal(res_b["ali"], res_f["ali"])
    us.close(); tm.close(); dm.close()
