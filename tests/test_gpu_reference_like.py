"""The behaviours the reference's own binding tests exercise (python/tests/test_diag_gmm.py, test_mle_diag_gmm.py,
test_am_diag_gmm.py, test_mle_am_diag_gmm.py), checked on this package's C++ host classes with fresh random values: same
method names, same return conventions, expected values from plain numpy formulas.  Likelihoods and posteriors run on the GPU."""
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def khg(ctx):
    import kaldi_hmm_gmm_amd as k
    from kaldi_hmm_gmm_amd import _gpu
    _gpu.set_default_context(ctx)
    return k


def _gmm(khg, rng, nmix, dim):
    g = khg.DiagGmm(nmix=nmix, dim=dim)
    w = rng.random(nmix).astype(np.float32); w /= w.sum()
    mean = rng.standard_normal((nmix, dim)).astype(np.float32)
    var = (rng.random((nmix, dim)) + 0.5).astype(np.float32)
    g.set_weights(w); g.set_means(mean); g.set_invvars(1 / var)
    return g, w, mean, var


def _loglikes(w, mean, var, x):
    """log(w_g N(x; mean_g, diag var_g)) per component, float64"""
    w, mean, var, x = (np.asarray(a, np.float64) for a in (w, mean, var, x))
    return np.log(w) - 0.5 * (np.log(2 * np.pi * var).sum(1) + ((x - mean) ** 2 / var).sum(1))


def test_get_set_remove(khg):
    rng = np.random.default_rng(1)
    nmix, dim = 10, 8
    g, w, mean, var = _gmm(khg, rng, nmix, dim)
    assert np.allclose(g.weights, w) and np.allclose(g.means, mean, atol=1e-6) and np.allclose(g.vars, var, rtol=1e-6)
    assert g.num_gauss == nmix and g.dim == dim and g.valid_gconsts is False
    assert g.compute_gconsts() == 0 and g.valid_gconsts is True
    want = np.log(w) - 0.5 * (dim * np.log(2 * np.pi) + np.log(var).sum(1) + (mean ** 2 / var).sum(1))
    assert np.allclose(g.gconsts, want, rtol=1e-5, atol=1e-5)
    assert np.allclose(g.means_invvars, mean / var, rtol=1e-6, atol=1e-6) and np.allclose(g.inv_vars, 1 / var, rtol=1e-6)
    for i in range(nmix):
        g.set_component_weight(i, float(w[i]))
    assert g.valid_gconsts is False and g.compute_gconsts() == 0
    for i in range(nmix):
        assert g.weights[i] == w[i]
        g.set_component_mean(i, mean[i]); g.set_component_inv_var(i, 1 / var[i])
        assert np.allclose(g.get_component_mean(i), mean[i], atol=1e-6) and np.allclose(g.get_component_variance(i), var[i], rtol=1e-6)
    g.set_invvars_and_means(1 / var, mean)
    assert np.allclose(g.means, mean, atol=1e-6) and np.allclose(g.vars, var, rtol=1e-6)
    g.remove_component(0, renorm_weights=True)
    assert g.num_gauss == nmix - 1 and g.dim == dim and np.allclose(g.weights, w[1:] / w[1:].sum(), rtol=1e-6)
    assert np.allclose(g.means, mean[1:], atol=1e-6)
    g.remove_component(1, renorm_weights=False)              # original component 2 goes, weights untouched
    keep = [1] + list(range(3, nmix))
    assert g.num_gauss == nmix - 2 and np.allclose(g.weights, (w[1:] / w[1:].sum())[[0] + list(range(2, nmix - 1))], rtol=1e-6)
    assert np.allclose(g.get_component_mean(0), mean[1], atol=1e-6) and np.allclose(g.means, mean[keep], atol=1e-6)
    g.remove_components([2, 1, 0], renorm_weights=True)
    assert g.num_gauss == nmix - 5 and np.allclose(g.means, mean[keep[3:]], atol=1e-6) and abs(g.weights.sum() - 1) < 1e-6
    g.remove_component(g.num_gauss - 1, renorm_weights=True)
    assert g.num_gauss == nmix - 6 and np.allclose(g.vars, var[keep[3:-1]], rtol=1e-6)
    with pytest.raises(khg.KhgError):
        one = khg.DiagGmm(nmix=1, dim=2); one.remove_component(0, True)


def test_split_returns_the_history(khg):
    rng = np.random.default_rng(2)
    g, w, mean, var = _gmm(khg, rng, 1, 5)
    new2old = g.split(target_components=2, perturb_factor=0.01)
    assert g.num_gauss == 2 and new2old == [0]              # the second component is split off the first
    assert g.weights[0] == g.weights[1] == w[0] / 2
    assert np.allclose(g.vars[0], var[0], rtol=1e-6) and np.allclose(g.vars[1], var[0], rtol=1e-6)
    assert np.allclose(g.means.sum(0), mean[0] * 2, atol=1e-5)      # +- the same perturbation
    g, w, mean, var = _gmm(khg, rng, 2, 5)
    g.set_weights(np.array([0.4, 0.6], np.float32))
    new2old = g.split(target_components=4, perturb_factor=0.01)
    assert g.num_gauss == 4 and new2old == [1, 0]           # the heaviest first: 0.6 -> 0.3 + 0.3, then 0.4 -> 0.2 + 0.2
    assert np.allclose(g.weights, [0.2, 0.3, 0.3, 0.2])
    assert np.allclose(g.vars[[0, 3]], var[[0, 0]], rtol=1e-6) and np.allclose(g.vars[[1, 2]], var[[1, 1]], rtol=1e-6)
    assert g.split(target_components=4, perturb_factor=0.01) in ([], None)      # nothing to do


def test_merge(khg):
    rng = np.random.default_rng(3)
    g, w, mean, var = _gmm(khg, rng, 4, 6)
    history = g.merge(target_components=1)
    assert history == [] and g.num_gauss == 1 and abs(g.weights[0] - 1) < 1e-6
    m = (w[:, None] * mean).sum(0)
    assert np.allclose(g.means[0], m, atol=1e-5)
    assert np.allclose(g.vars[0], (w[:, None] * (var + mean ** 2)).sum(0) - m ** 2, rtol=1e-4, atol=1e-5)
    # two near-identical components among four: they are the pair that merges
    g, w, mean, var = _gmm(khg, rng, 4, 6)
    mean[2] = mean[0] + 1e-3; var[2] = var[0]
    g.set_means(mean); g.set_invvars(1 / var); g.set_means(mean)
    history = g.merge(target_components=3)
    assert history == [2, 0] and g.num_gauss == 3
    assert g.weights[0] == w[1] and g.weights[1] == w[2] + w[0] and g.weights[2] == w[3]
    assert np.allclose(g.means[0], mean[1], atol=1e-6) and np.allclose(g.means[2], mean[3], atol=1e-6)
    assert np.allclose(g.means[1], (w[0] * mean[0] + w[2] * mean[2]) / (w[0] + w[2]), atol=1e-5)


def test_log_likes_per_component_2d_preselect_and_posteriors(khg):
    rng = np.random.default_rng(4)
    nmix, dim, N = 9, 7, 5
    g, w, mean, var = _gmm(khg, rng, nmix, dim)
    g.compute_gconsts()
    x = rng.standard_normal(dim).astype(np.float32)
    comp = _loglikes(w, mean, var, x)
    lse = lambda v: float(np.max(v) + np.log(np.exp(v - np.max(v)).sum()))      # noqa: E731
    assert abs(g.log_likelihood(x) - lse(comp)) < 1e-4
    assert np.allclose(g.log_likelihoods(x), comp, atol=1e-4)
    X = rng.standard_normal((N, dim)).astype(np.float32)
    mat = g.log_likelihoods_matrix(X)
    assert mat.shape == (N, nmix) and np.allclose(mat, np.stack([_loglikes(w, mean, var, r) for r in X]), atol=1e-4)
    idx = [7, 0, 3]
    pre = g.log_likelihoods_preselect(x, idx)
    assert pre.shape[0] == len(idx) and np.allclose(pre, comp[idx], atol=1e-4)
    for i in range(nmix):
        assert abs(g.component_log_likelihood(x, i) - comp[i]) < 1e-4
    ll, post = g.component_posteriors(x)
    assert np.allclose(post, np.exp(comp - lse(comp)), atol=1e-5) and abs(ll - lse(comp)) < 1e-4
    # Gaussian selection: the best components, best first, and the log-sum of their likelihoods
    order = np.argsort(-comp)
    ll3, sel = g.gaussian_selection_1d(x, 3)
    assert sel == order[:3].tolist() and abs(ll3 - lse(comp[order[:3]])) < 1e-4
    tot, sel2 = g.gaussian_selection_2d(X, 2)
    want = 0.0
    for i, r in enumerate(X):
        c = _loglikes(w, mean, var, r); o = np.argsort(-c)
        assert sel2[i] == o[:2].tolist()
        want += lse(c[o[:2]])
    assert abs(tot - want) < 1e-3
    preselect = [8, 2, 5, 1, 6]
    llp, selp = g.gaussian_selection_preselect(x, preselect, 2)
    po = sorted(preselect, key=lambda k: -comp[k])
    assert selp == po[:2] and abs(llp - lse(comp[po[:2]])) < 1e-4
    llq, selq = g.gaussian_selection_1d(x, nmix + 5)          # more than there are: all of them, best first
    assert selq == order.tolist() and abs(llq - lse(comp)) < 1e-4


def test_generate_perturb_copy_interpolate_pickle(khg):
    rng = np.random.default_rng(5)
    nmix, dim = 3, 4
    g, w, mean, var = _gmm(khg, rng, nmix, dim)
    g.set_invvars(np.full((nmix, dim), 1e6, np.float32)); g.set_means(mean)      # tiny variances: a sample sits on a mean
    x = g.generate()
    assert x.shape == (dim,) and min(np.abs(x - m).max() for m in mean) < 0.05
    g, w, mean, var = _gmm(khg, rng, nmix, dim)
    g.compute_gconsts()
    g.perturb(0.1)
    assert np.allclose(g.vars, var, rtol=1e-6) and not np.allclose(g.means, mean) and g.valid_gconsts
    c = khg.DiagGmm(); c.copy_from_diag_gmm(g)
    assert c.valid_gconsts == g.valid_gconsts and np.array_equal(c.gconsts, g.gconsts) and np.array_equal(c.means_invvars, g.means_invvars)
    p = pickle.loads(pickle.dumps(g))
    assert np.array_equal(p.weights, g.weights) and np.array_equal(p.inv_vars, g.inv_vars) and p.valid_gconsts
    other, w2, mean2, var2 = _gmm(khg, rng, nmix, dim)
    before = g.means.copy()
    g.interpolate(0.25, other)
    assert np.allclose(g.means, 0.75 * before + 0.25 * mean2, atol=1e-5) and abs(g.weights.sum() - 1) < 1e-6


def test_accum_diag_gmm_like_reference(khg):
    num_gauss, dim = 3, 5
    acc = khg.AccumDiagGmm()
    acc.resize(num_gauss=num_gauss, dim=dim, flags=khg.GmmUpdateFlags.kGmmAll)
    assert acc.flags == khg.GmmUpdateFlags.kGmmAll and acc.num_gauss == num_gauss and acc.dim == dim
    assert acc.occupancy.shape == (num_gauss,) and acc.mean_accumulator.shape == acc.variance_accumulator.shape == (num_gauss, dim)
    assert acc.occupancy.dtype == acc.mean_accumulator.dtype == acc.variance_accumulator.dtype == np.float64
    acc.resize(num_gauss=num_gauss, dim=dim, flags=khg.GmmUpdateFlags.kGmmWeights)
    assert acc.flags == khg.GmmUpdateFlags.kGmmWeights and len(acc.mean_accumulator) == 0 and len(acc.variance_accumulator) == 0
    acc.resize(num_gauss=num_gauss, dim=dim, flags=khg.GmmUpdateFlags.kGmmMeans)
    assert acc.mean_accumulator.shape == (num_gauss, dim) and len(acc.variance_accumulator) == 0
    acc.resize(num_gauss=num_gauss, dim=dim, flags=khg.GmmUpdateFlags.kGmmVariances)       # variances need the means
    assert acc.mean_accumulator.shape == acc.variance_accumulator.shape == (num_gauss, dim)
    rng = np.random.default_rng(6)
    acc.resize(num_gauss=num_gauss, dim=dim, flags=khg.GmmUpdateFlags.kGmmAll)
    d = rng.standard_normal(dim).astype(np.float32)
    acc.accumulate_for_component(data=d, comp_index=1, weight=0.25)
    assert np.allclose(acc.occupancy, [0, 0.25, 0]) and np.allclose(acc.mean_accumulator[1], d.astype(np.float64) * 0.25)
    assert np.allclose(acc.variance_accumulator[1], (d.astype(np.float64) ** 2) * 0.25)
    post = np.array([0.1, 0.6, 0.3], np.float32)
    occ, ma, va = acc.occupancy.copy(), acc.mean_accumulator.copy(), acc.variance_accumulator.copy()
    acc.accumulate_from_posteriors(data=d, gauss_posteriors=post)
    assert np.allclose(acc.occupancy, occ + post) and np.allclose(acc.mean_accumulator, ma + np.outer(post, d))
    assert np.allclose(acc.variance_accumulator, va + np.outer(post, d * d))
    acc.scale(f=0.1, flags=khg.GmmUpdateFlags.kGmmAll)
    assert np.allclose(acc.occupancy, (occ + post) * 0.1)
    acc.set_zero(khg.GmmUpdateFlags.kGmmAll)
    assert acc.occupancy.sum() == 0 and np.abs(acc.mean_accumulator).sum() == 0 and np.abs(acc.variance_accumulator).sum() == 0


def test_accumulate_from_diag_and_update_like_reference(khg):
    """accumulate_from_diag = posteriors of the frame under the model times the weight (on the GPU), then mle_diag_gmm_update moves
    the model towards the data: the objective change is positive and the count is the total occupancy."""
    rng = np.random.default_rng(7)
    nmix, dim = 4, 6
    g, w, mean, var = _gmm(khg, rng, nmix, dim)
    g.compute_gconsts()
    acc = khg.AccumDiagGmm(g, khg.GmmUpdateFlags.kGmmAll)
    data = (mean[rng.integers(0, nmix, 400)] + 0.3 + rng.standard_normal((400, dim)) * np.sqrt(var.mean())).astype(np.float32)
    tot = 0.0
    for x in data[:50]:
        comp = _loglikes(w, mean, var, x)
        ll = acc.accumulate_from_diag(g, x, 0.5)
        assert abs(ll - float(np.max(comp) + np.log(np.exp(comp - np.max(comp)).sum()))) < 1e-3
        tot += 0.5
    assert abs(acc.occupancy.sum() - tot) < 1e-4
    for x in data[50:]:
        acc.accumulate_from_diag(g, x, 1.0)
    obj, count, floored_elems, floored_gauss, removed = khg.mle_diag_gmm_update(khg.MleDiagGmmOptions(), acc, khg.GmmUpdateFlags.kGmmAll, g)
    assert obj > 0 and abs(count - (tot + 350)) < 1e-2 and removed == 0 and g.valid_gconsts
    assert np.abs(g.means - mean).max() > 0.05


def test_am_diag_gmm_and_accum_am_like_reference(khg):
    rng = np.random.default_rng(8)
    dim = 5
    am = khg.AmDiagGmm()
    gs = []
    for nmix in (2, 3, 4):
        g, *_ = _gmm(khg, rng, nmix, dim)
        g.compute_gconsts()
        am.add_pdf(g); gs.append(g)
    assert am.num_pdfs == 3 and am.dim == dim and am.num_gauss == 9 and am.num_gauss_in_pdf(1) == 3
    x = rng.standard_normal(dim).astype(np.float32)
    for i, g in enumerate(gs):
        assert abs(am.log_likelihood(i, x) - g.log_likelihood(x)) < 1e-6
    am.get_pdf(0).set_component_weight(0, 0.9)                     # get_pdf returns a reference ...
    assert am.get_pdf(0).weights[0] == np.float32(0.9) and gs[0].weights[0] != np.float32(0.9)      # ... add_pdf had copied
    am.compute_gconsts()
    accs = khg.AccumAmDiagGmm()
    accs.init(am, khg.GmmUpdateFlags.kGmmAll)
    assert accs.num_accs == 3 and accs.dim == dim
    ll = accs.accumulate_for_gmm(am, x, 2, 1.0)
    assert abs(ll - am.log_likelihood(2, x)) < 1e-4 and abs(accs.tot_count - 1.0) < 1e-6 and abs(accs.tot_log_like - ll) < 1e-4
    a2 = accs.get_acc(2)                                          # a copy
    a2.occupancy[:] = 0
    assert abs(accs.get_acc(2).occupancy.sum() - 1.0) < 1e-5 and abs(accs.tot_stats_count - 1.0) < 1e-5
    other = khg.AccumAmDiagGmm(); other.init(am, khg.GmmUpdateFlags.kGmmAll)
    other.add(2.0, accs)
    assert abs(other.tot_count - 2.0) < 1e-6 and abs(other.get_acc(2).occupancy.sum() - 2.0) < 1e-5
    am2 = pickle.loads(pickle.dumps(am))
    assert am2.num_pdfs == 3 and np.array_equal(am2.get_pdf(1).means_invvars, am.get_pdf(1).means_invvars)
