"""K4: the device M-step (khg_model_mle_update) against the oracle's restatement of
MleDiagGmmUpdate (csrc/mle-diag-gmm.cc:243-390) and against the host M-step (khg_mle_am_diag_gmm_update).

Stated tolerance: weights / inv_vars / means_invvars / gauss_off / count / floored / removed are BIT-EXACT
(IEEE fp64 and fp32 operations in the reference's order, contraction off).  gconsts go through logf, where the
device's libm and glibc may round differently in the last place: |delta| <= 4 float ulps.  objf_change is a
float difference of two ~1e5-sized sums: |delta| <= 2e-6 * (sum |occ * gconst|)."""
import numpy as np
import pytest

from helpers import build
from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, MleDiagGmmOptions, synth
from kaldi_hmm_gmm_amd import mle as khg_mle
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GC_ULPS = 4      # glibc's and the device's logf are each within ~1 ulp of the exact value: up to 2 ulps apart per log, a few in the sum (observed: <= 2 at D >= 13, 3 at D = 1)


def _fake_accs(m, rng, frames_per_gauss=40.0):
    """Accumulators as K3 would leave them for data drawn near (not at) the model, plus the corner cases of
    the update: low-occupancy components (removed), a pdf where all but one are low (the last one is kept with a
    floored weight), zero-variance statistics (floored), an all-zero pdf (occ_sum == 0)."""
    G, D = int(m.gauss_off[-1]), m.means.shape[1]
    occ = rng.uniform(0.5, 2.0, G) * frames_per_gauss
    mu = m.means.astype(np.float64) + 0.3 * rng.standard_normal((G, D))
    var = m.vars.astype(np.float64) * rng.uniform(0.7, 1.4, (G, D))
    P = len(m.gauss_off) - 1
    for p in range(P):
        a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        if b - a >= 3 and p % 3 == 0:
            occ[a + 1] = 2.0                      # < min_gaussian_occupancy
        if b - a >= 2 and p % 5 == 1:
            var[a] = 0.0                          # zero variance -> floored
        if p % 7 == 2:
            occ[a:b] = 1.5                        # every component low: G-1 removed, the last weight-floored
        if p % 11 == 5:
            occ[a:b] = 0.0                        # no data at all
        if b - a >= 4 and p % 4 == 3:
            occ[a + 2] = 1e-7 * occ[a:b].sum()    # weight below min_gaussian_weight only when occupancy is high
    mean_acc = occ[:, None] * mu
    var_acc = occ[:, None] * (var + mu * mu)
    return occ, mean_acc, var_acc


def _upload(accs, occ, mean_acc, var_acc):
    buf = np.zeros(accs.size, np.float64)
    G, D = accs.sumG, accs.dim
    buf[:G] = occ
    buf[G: G + G * D] = mean_acc.reshape(-1)
    buf[G + G * D: G + 2 * G * D] = var_acc.reshape(-1)
    accs.upload(buf)


def _ulps(a, b):
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))


@pytest.mark.parametrize("flags", ["mvw", "mw", "w", "v", "m", "mv"])
@pytest.mark.parametrize("shape", [(23, 7, 13, False), (40, 12, 40, True), (6, 70, 80, True)])
def test_device_m_step_vs_oracle_and_host(ctx, flags, shape):
    P, Gmax, D, ragged = shape
    rng = np.random.default_rng(17 + P)
    m = synth.make_model(P, Gmax, D, seed=31 + P, ragged=ragged)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    occ, mean_acc, var_acc = _fake_accs(m, rng)
    f = int(khg_mle.str_to_gmm_flags(flags))
    opts = MleDiagGmmOptions()
    # oracle, pdf by pdf
    ref = []
    for p in range(P):
        a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        ref.append(orc.mle_diag_gmm_update(m.weights[a:b], m.means_invvars[a:b], m.inv_vars[a:b], occ[a:b], mean_acc[a:b],
                                           var_acc[a:b], acc_flags=0xF, flags=f))
    # host product M-step
    h_off, h_w, h_gc, h_miv, h_iv, h_obj, h_cnt, h_fe, h_fg, h_rm = khg_mle._flat_update(
        opts, m.gauss_off, occ, mean_acc, var_acc, 0x7, f, m.weights, m.means_invvars, m.inv_vars)
    # device
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    tm = DeviceTransitions(ctx, m.id2pdf)
    accs = DeviceAccs(ctx, dm, tm)
    _upload(accs, occ, mean_acc, var_acc)
    r = dm.mle_update(accs, opts, f)
    d = dm.download()
    assert np.array_equal(d["gauss_off"], h_off)
    assert r["removed"] == h_rm == sum(x["removed"] for x in ref) and r["removed"] > 0
    assert r["floored_elements"] == h_fe and r["floored_gaussians"] == h_fg
    if f & 2:
        assert r["floored_elements"] > 0
    assert r["count"] == h_cnt
    for name, got, want in (("weights", d["weights"], h_w), ("inv_vars", d["inv_vars"], h_iv),
                            ("means_invvars", d["means_invvars"], h_miv)):
        np.testing.assert_array_equal(got, want, err_msg=name)
    tot = np.float32(0)
    scale = 0.0
    for p in range(P):
        a, b = int(h_off[p]), int(h_off[p + 1])
        for name in ("weights", "inv_vars", "means_invvars"):
            np.testing.assert_array_equal(d[name][a:b], ref[p][name], err_msg=f"pdf {p} {name} vs oracle")
        u = _ulps(d["gconsts"][a:b], ref[p]["gconsts"])
        assert u.max() <= GC_ULPS, (p, u.max())
        tot = np.float32(tot + np.float32(ref[p]["obj_change"]))
        a0, b0 = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        scale += float(np.abs(occ[a0:b0] * gc[a0:b0]).sum()) + float(np.abs(mean_acc[a0:b0] * m.means_invvars[a0:b0]).sum())
    assert _ulps(d["gconsts"], h_gc).max() <= GC_ULPS
    assert (d["gconsts"] == h_gc).mean() > 0.9          # the logf difference is rare
    assert abs(r["objf_change"] - float(tot)) <= 2e-6 * scale + 1e-3
    assert abs(r["objf_change"] - h_obj) <= 2e-6 * scale + 1e-3


def test_device_m_step_model_is_repacked_for_k1_k3(ctx):
    """After the update (with removals) the handle must behave exactly like a model created from the same
    parameters: K1 log-likes and K3 statistics through the updated handle == through a fresh handle."""
    m, gc, om, ut, cost = build(30, 9, 20, 12, seed=4, ragged=True, max_phones=5)
    rng = np.random.default_rng(3)
    occ, mean_acc, var_acc = _fake_accs(m, rng)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    accs = DeviceAccs(ctx, dm, tm)
    _upload(accs, occ, mean_acc, var_acc)
    objf, count = khg_mle.mle_am_diag_gmm_update_device(MleDiagGmmOptions(), accs, 0x7, dm)
    d = dm.download()
    assert d["gauss_off"][-1] < m.gauss_off[-1] and accs.sumG == d["gauss_off"][-1]
    fresh = DeviceModel(ctx, d["gauss_off"], d["gconsts"], d["means_invvars"], d["inv_vars"])
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    us.loglikes(dm)
    ll_a = us.download_loglikes()
    us.loglikes(fresh)
    ll_b = us.download_loglikes()
    for x, y in zip(ll_a, ll_b):
        np.testing.assert_array_equal(x, y)
    us.upload_ali(ut.ref_ali)
    accs_b = DeviceAccs(ctx, fresh, tm)
    us.acc_stats(dm, tm, accs)
    us.acc_stats(fresh, tm, accs_b)
    sa, sb = accs.download(), accs_b.download()
    np.testing.assert_allclose(sa["occ"], sb["occ"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(sa["mean_acc"], sb["mean_acc"], rtol=1e-9, atol=1e-9)
    tr = accs.download_trans()
    assert np.array_equal(tr["trans_acc"], sa["trans_acc"]) and tr["total_frames"] == sa["total_frames"]


def test_device_m_step_two_em_iterations_match_host_path(ctx):
    """align -> acc-stats -> M-step twice: the all-device loop and the host-M-step loop stay together
    (same alignments; parameters bit-equal except gconsts within GC_ULPS)."""
    m, gc, om, ut, cost = build(24, 4, 16, 40, seed=9, max_phones=5)
    opts = MleDiagGmmOptions(min_gaussian_occupancy=3.0)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    accs = DeviceAccs(ctx, dm, tm)
    go, w, miv, iv, g = m.gauss_off, m.weights, m.means_invvars, m.inv_vars, gc
    for it in range(2):
        # host-M-step path on a fresh handle
        hm = DeviceModel(ctx, go, g, miv, iv)
        haccs = DeviceAccs(ctx, hm, tm)
        us.loglikes(hm, reachable_only=True)
        ra = us.align(tm, beam=200.0, acoustic_scale=0.1)
        us.acc_stats(hm, tm, haccs)
        st = haccs.download()
        go2, w2, g2, miv2, iv2, *_ = khg_mle._flat_update(opts, go, st["occ"], st["mean_acc"], st["var_acc"], 0x7, 0x7, w, miv, iv)
        # all-device path
        accs.zero()
        us.loglikes(dm, reachable_only=True)
        rb = us.align(tm, beam=200.0, acoustic_scale=0.1)
        us.acc_stats(dm, tm, accs)
        assert np.array_equal(ra["ali"], rb["ali"]) and not np.any(ra["status"] & 1)     # bit 0 = ALIGN_ERROR
        khg_mle.mle_am_diag_gmm_update_device(opts, accs, 0x7, dm)
        d = dm.download()
        assert np.array_equal(d["gauss_off"], go2)
        # K3's fp64 atomics land in a run-dependent order, so the two accumulator sets agree to ~1e-13
        # relative, not bitwise; parameters therefore agree to float rounding of that
        np.testing.assert_allclose(d["weights"], w2, rtol=1e-6)
        np.testing.assert_allclose(d["inv_vars"], iv2, rtol=1e-5)
        np.testing.assert_allclose(d["means_invvars"], miv2, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d["gconsts"], g2, rtol=1e-5, atol=1e-5)
        go, w, g, miv, iv = go2, w2.copy(), g2.copy(), miv2.copy(), iv2.copy()


def test_device_m_step_fixed_point_at_full_size(ctx):
    """Size-independent property at BASELINE's 5000 x 64 x 40: statistics that ARE the model's own moments
    (occ = N w, mean_acc = occ mu, var_acc = occ (var + mu^2)) give the model back (float rounding only),
    remove nothing and floor nothing."""
    P, G, D = 5000, 64, 40
    m = synth.make_model(P, G, D, seed=20230418)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    tm = DeviceTransitions(ctx, m.id2pdf)
    accs = DeviceAccs(ctx, dm, tm)
    mu = (m.means_invvars.astype(np.float64) / m.inv_vars.astype(np.float64))
    var = 1.0 / m.inv_vars.astype(np.float64)
    occ = 6000.0 * m.weights.astype(np.float64)
    _upload(accs, occ, occ[:, None] * mu, occ[:, None] * (var + mu * mu))
    r = dm.mle_update(accs, MleDiagGmmOptions(), 0x7)
    assert r["removed"] == 0 and r["floored_elements"] == 0
    assert r["count"] == pytest.approx(6000.0 * P, rel=1e-6)
    d = dm.download()
    np.testing.assert_allclose(d["weights"], m.weights, rtol=2e-6)
    np.testing.assert_allclose(d["inv_vars"], m.inv_vars, rtol=1e-4)     # var = E[x^2] - mu^2 cancels ~1e-12 * 10
    np.testing.assert_allclose(d["means_invvars"], m.means_invvars, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d["gconsts"], gc, rtol=1e-5, atol=1e-4)
    assert abs(r["objf_change"]) <= 1e-7 * 6000.0 * P * 100


def test_device_m_step_variance_floor_vector(ctx):
    """MleDiagGmmOptions.variance_floor_vector (csrc/mle-diag-gmm.cc:311-322) through K4: per-dimension floors instead of
    min_variance; parameters bit-exact vs the host M-step and the oracle, the same floored counters."""
    P, Gmax, D = 17, 9, 23
    rng = np.random.default_rng(91)
    m = synth.make_model(P, Gmax, D, seed=77, ragged=True)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    occ, mean_acc, var_acc = _fake_accs(m, rng)
    floor = rng.uniform(1e-3, 2.5, D)                  # the synthetic variances are U[0.35, 2.8]: a good share gets floored
    opts = MleDiagGmmOptions(min_variance=1e9, variance_floor_vector=floor)     # min_variance must be ignored
    h_off, h_w, h_gc, h_miv, h_iv, h_obj, h_cnt, h_fe, h_fg, h_rm = khg_mle._flat_update(
        opts, m.gauss_off, occ, mean_acc, var_acc, 0x7, 7, m.weights, m.means_invvars, m.inv_vars)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    tm = DeviceTransitions(ctx, m.id2pdf)
    accs = DeviceAccs(ctx, dm, tm)
    _upload(accs, occ, mean_acc, var_acc)
    r = dm.mle_update(accs, opts, 7)
    d = dm.download()
    assert np.array_equal(d["gauss_off"], h_off) and r["removed"] == h_rm
    assert r["floored_elements"] == h_fe > 100 and r["floored_gaussians"] == h_fg
    assert float((1.0 / d["inv_vars"]).max()) < 1e6     # nothing was floored at min_variance = 1e9
    for name, want in (("weights", h_w), ("inv_vars", h_iv), ("means_invvars", h_miv)):
        np.testing.assert_array_equal(d[name], want, err_msg=name)
    for p in range(P):
        a0, b0 = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        ref = orc.mle_diag_gmm_update(m.weights[a0:b0], m.means_invvars[a0:b0], m.inv_vars[a0:b0], occ[a0:b0], mean_acc[a0:b0],
                                      var_acc[a0:b0], acc_flags=0xF, flags=7, min_variance=1e9, variance_floor_vector=floor)
        a, b = int(h_off[p]), int(h_off[p + 1])
        for name in ("weights", "inv_vars", "means_invvars"):
            np.testing.assert_array_equal(d[name][a:b], ref[name], err_msg=f"pdf {p} {name} vs oracle")
    assert _ulps(d["gconsts"], h_gc).max() <= GC_ULPS


def test_device_split_equals_host_split(ctx):
    """khg_model_split (DiagGmm::Split per pdf on the device, injected normal deviates) == the host form (DiagGmm.split,
    csrc/diag-gmm.cc:780-851 restated) fed the same deviates: gauss_off, weights, inv_vars, means_invvars bit for bit,
    gconsts within the logf tolerance; a pdf whose target equals its size is left alone; a smaller target is an error."""
    import kaldi_hmm_gmm_amd as khg

    P, Gmax, D = 11, 6, 13
    rng = np.random.default_rng(5)
    m = synth.make_model(P, Gmax, D, seed=12, ragged=True)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    cur = np.diff(m.gauss_off)
    tgt = (cur + rng.integers(0, 5, P)).astype(np.int32)
    tgt[3] = cur[3]
    n_new = int((tgt - cur).sum())
    normals = rng.standard_normal((n_new, D)).astype(np.float32)
    # host form, pdf by pdf, consuming the same stream
    it = iter(normals)
    want_w, want_miv, want_iv, want_gc = [], [], [], []
    for p in range(P):
        a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        g = khg.DiagGmm(nmix=b - a, dim=D)
        g.set_weights(m.weights[a:b]); g.set_invvars(m.inv_vars[a:b]); g._means_invvars = m.means_invvars[a:b].copy(); g.compute_gconsts()
        g.split(int(tgt[p]), 0.01, randn=lambda d: next(it))
        want_w.append(g.weights); want_miv.append(g.means_invvars); want_iv.append(g.inv_vars); want_gc.append(g.gconsts)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    dm.split(tgt, 0.01, normals)
    d = dm.download()
    assert np.array_equal(d["gauss_off"], np.concatenate([[0], np.cumsum(tgt)]).astype(np.int32))
    np.testing.assert_array_equal(d["weights"], np.concatenate(want_w))
    np.testing.assert_array_equal(d["inv_vars"], np.concatenate(want_iv))
    np.testing.assert_array_equal(d["means_invvars"], np.concatenate(want_miv))
    assert _ulps(d["gconsts"], np.concatenate(want_gc)).max() <= GC_ULPS
    # the split model scores: K1 on the new tile image gives finite log-likes
    tm = DeviceTransitions(ctx, m.id2pdf)
    us = UtteranceSet(ctx, None, np.array([0, 40], np.int64), rng.standard_normal((40, D)).astype(np.float32))
    us.set_pdf_list(np.arange(P, dtype=np.int32))
    us.loglikes(dm)
    assert np.isfinite(us.download_loglikes()[0]).all()
    from kaldi_hmm_gmm_amd import KhgError
    with pytest.raises(KhgError, match="Cannot split"):
        dm.split(np.maximum(tgt - 1, 1).astype(np.int32), 0.01, normals)
    for o in (us, tm, dm):
        o.close()


@pytest.mark.parametrize("shape", [(11, 24, 13), (4, 70, 40)])
def test_device_merge_equals_host_merge(ctx, shape):
    """khg_model_merge (DiagGmm::Merge per pdf on the device, csrc/diag-gmm.cc:557-759) == the host form
    (khg_diag_gmm_merge through DiagGmm.merge): same component counts; weights / inv_vars / means_invvars to 2e-5 relative
    (the pair costs go through the device logf, <= 2 ulp from glibc's, the merged parameters themselves through the same
    float operations), gconsts to 1e-4 absolute + 1e-5 relative; targets of 1 (the global-moments branch), of the current
    size (left alone, bit for bit) and in between; invalid targets are errors with the reference's wording."""
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd import KhgError

    P, Gmax, D = shape
    rng = np.random.default_rng(17)
    m = synth.make_model(P, Gmax, D, seed=21, ragged=True)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    cur = np.diff(m.gauss_off)
    tgt = np.maximum(1, cur - rng.integers(0, np.maximum(cur, 2) - 1)).astype(np.int32)
    tgt[0] = 1
    tgt[1] = cur[1]
    if P > 2:
        tgt[2] = max(1, cur[2] // 2)
    want_w, want_miv, want_iv, want_gc = [], [], [], []
    for p in range(P):
        a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        g = khg.DiagGmm(nmix=b - a, dim=D)
        g.set_weights(m.weights[a:b]); g.set_invvars(m.inv_vars[a:b]); g._means_invvars = m.means_invvars[a:b].copy(); g.compute_gconsts()
        if tgt[p] < b - a:
            g.merge(int(tgt[p]))
            want_gc.append(g.gconsts)
        else:
            want_gc.append(gc[a:b])
        assert g.num_gauss == tgt[p]
        want_w.append(g.weights); want_miv.append(g.means_invvars); want_iv.append(g.inv_vars)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    dm.merge(tgt)
    d = dm.download()
    assert np.array_equal(d["gauss_off"], np.concatenate([[0], np.cumsum(tgt)]).astype(np.int32))
    assert np.array_equal(np.asarray(dm.gauss_off), d["gauss_off"])
    np.testing.assert_allclose(d["weights"], np.concatenate(want_w), rtol=2e-5, atol=0)
    np.testing.assert_allclose(d["inv_vars"], np.concatenate(want_iv), rtol=2e-5, atol=0)
    np.testing.assert_allclose(d["means_invvars"], np.concatenate(want_miv), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(d["gconsts"], np.concatenate(want_gc), rtol=1e-5, atol=1e-4)
    a1, b1 = int(d["gauss_off"][1]), int(d["gauss_off"][2])          # the untouched pdf: bit for bit, gconsts included
    a0, b0 = int(m.gauss_off[1]), int(m.gauss_off[2])
    assert np.array_equal(d["weights"][a1:b1], m.weights[a0:b0]) and np.array_equal(d["gconsts"][a1:b1], gc[a0:b0])
    assert np.array_equal(d["means_invvars"][a1:b1], m.means_invvars[a0:b0])
    # the merged model scores on the re-packed image
    us = UtteranceSet(ctx, None, np.array([0, 40], np.int64), rng.standard_normal((40, D)).astype(np.float32))
    us.set_pdf_list(np.arange(P, dtype=np.int32))
    us.loglikes(dm)
    assert np.isfinite(us.download_loglikes()[0]).all()
    # merging twice the same way gives the same model (no dependence on scratch contents)
    dm2 = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    dm2.merge(tgt)
    d2 = dm2.download()
    for k in ("weights", "gconsts", "means_invvars", "inv_vars"):
        assert np.array_equal(d[k], d2[k])
    now = np.diff(d["gauss_off"])
    with pytest.raises(KhgError, match="Invalid argument for target number of Gaussians"):
        dm.merge((now + 1).astype(np.int32))
    with pytest.raises(KhgError, match="Invalid argument for target number of Gaussians"):
        dm.merge(np.zeros(P, np.int32))
    dm.merge(now.astype(np.int32))        # nothing to do
    assert np.array_equal(dm.download()["weights"], d["weights"])
