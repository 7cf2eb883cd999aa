"""CPU-side tests of the host logic behind the reference's names: no GPU needed.
Golden material comes from the reference's own tests (file:line cited)."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest

import kaldi_hmm_gmm_amd as khg
from kaldi_hmm_gmm_amd import _lib
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOPO_4PHONES = """
 <Topology>
 <TopologyEntry>
 <ForPhones> 1 </ForPhones>
 <State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 </State>
 <State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
 <State> 2 <PdfClass> 2 <Transition> 2 0.5 <Transition> 3 0.5 </State>
 <State> 3 <PdfClass> 3 <Transition> 3 0.5 <Transition> 4 0.5 </State>
 <State> 4 <PdfClass> 4 <Transition> 4 0.5 <Transition> 5 0.5 </State>
 <State> 5 </State>
 </TopologyEntry>
 <TopologyEntry>
 <ForPhones> 2 3 4 </ForPhones>
 <State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 </State>
 <State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
 <State> 2 <PdfClass> 2 <Transition> 2 0.5 <Transition> 3 0.5 </State>
 <State> 3 </State>
 </TopologyEntry>
 </Topology>
"""


def _tm():
    topo = khg.HmmTopology()
    topo.read(TOPO_4PHONES)
    tree = khg.monophone_context_dependency(phones=topo.phones, phone2num_pdf_classes=topo.get_phone_to_num_pdf_classes())
    return topo, tree, khg.TransitionModel(ctx_dep=tree, hmm_topo=topo)


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/khg_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "khg_hip.h")).read()
    declared = set(re.findall(r"\b(khg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in khg_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.khg_version() == 100


def test_no_gpu_fails_loudly():
    """Without a HIP device the product refuses to run instead of falling back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(khg.KhgError):
        khg.Context(0)


def test_hmm_topology_reference_example():
    # python/tests/test_hmm_topology.py:13-70
    s = """<Topology> <TopologyEntry> <ForPhones> 1 3 </ForPhones>
    <State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 </State>
    <State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
    <State> 2 <PdfClass> 2 <Transition> 2 0.5 <Transition> 3 0.5 </State>
    <State> 3 </State> </TopologyEntry> </Topology>"""
    topo = khg.HmmTopology()
    topo.read(s)
    assert topo.is_hmm is True
    assert topo.phones == [1, 3]
    assert topo.get_phone_to_num_pdf_classes() == [-1, 3, -1, 3]
    pt = topo.topology_for_phone(1)
    assert pt[0].forward_pdf_class == 0 and pt[0].self_loop_pdf_class == 0
    assert pt[0].transitions == [(0, 0.5), (1, 0.5)]
    assert pt[3].forward_pdf_class == -1 and pt[3].transitions == []
    assert topo.min_length(1) == 3
    topo2 = pickle.loads(pickle.dumps(topo, 2))
    assert str(topo2) == str(topo)
    with pytest.raises(khg.KhgError):
        topo.topology_for_phone(2)


def test_transition_model_golden():
    # python/tests/test_transition_model.py:82-142 and the text dump :185-230
    topo, tree, tm = _tm()
    assert tm.phones == [1, 2, 3, 4]
    assert tm.transition_id_to_pdf_array() == [0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10,
                                               11, 11, 12, 12, 13, 13]
    assert tm.is_self_loop(1) and not tm.is_self_loop(2) and tm.is_self_loop(3) and not tm.is_self_loop(4)
    assert tm.transition_ids_equivalent(1, 2) and not tm.transition_ids_equivalent(1, 3)
    assert tm.transition_ids_is_start_of_phone(1) and tm.transition_ids_is_start_of_phone(2)
    assert not tm.transition_ids_is_start_of_phone(3)
    assert [tm.transition_id_to_phone(t) for t in (1, 2, 10, 11, 16, 17)] == [1, 1, 1, 2, 2, 3]
    assert tm.is_final(1) is False and tm.is_final(10) is True
    assert tm.num_pdfs == 5 + 3 * 3 and tree.num_pdfs == 14
    triples = [(t.phone, t.hmm_state, t.forward_pdf) for t in tm.tuples]
    assert triples == [(1, 0, 0), (1, 1, 1), (1, 2, 2), (1, 3, 3), (1, 4, 4), (2, 0, 5), (2, 1, 6), (2, 2, 7), (3, 0, 8),
                       (3, 1, 9), (3, 2, 10), (4, 0, 11), (4, 1, 12), (4, 2, 13)]
    assert "<Triples> 14" in str(tm)
    lp = np.asarray(tm.log_probs)
    assert lp[0] == 0 and np.allclose(lp[1:], -0.693147, atol=1e-6) and len(lp) == 29
    stats = tm.init_stats()
    assert stats.dtype == np.float64 and stats.shape[0] == tm.num_transition_ids + 1 and stats.sum() == 0
    stats = tm.accumulate(prob=0.25, trans_id=1, stats=stats)
    stats = tm.accumulate(prob=0.25, trans_id=1, stats=stats)
    stats = tm.accumulate(prob=1.0, trans_id=10, stats=stats)
    assert stats[1] == 0.5 and stats[10] == 1.0
    tm2 = pickle.loads(pickle.dumps(tm, 2))
    assert str(tm2.topo) == str(tm.topo) and tm2.state2id == tm.state2id and tm2.id2state == tm.id2state
    assert tm2.id2pdf_id == tm.id2pdf_id and tm2.log_probs == tm.log_probs and tm2.num_pdfs == tm.num_pdfs
    assert tm2.non_self_loop_log_probs == tm.non_self_loop_log_probs
    assert all(str(a) == str(b) for a, b in zip(tm.tuples, tm2.tuples))


def test_gmm_update_flags():
    # python/tests/test_gmm_update_flags.py:9-39
    assert int(khg.GmmUpdateFlags.kGmmMeans) == 1 and khg.gmm_flags_to_str(khg.GmmUpdateFlags.kGmmMeans) == "m"
    assert int(khg.GmmUpdateFlags.kGmmVariances) == 2 and int(khg.GmmUpdateFlags.kGmmWeights) == 4
    assert int(khg.GmmUpdateFlags.kGmmTransitions) == 8
    assert int(khg.GmmUpdateFlags.kGmmAll) == 15 and khg.gmm_flags_to_str(khg.GmmUpdateFlags.kGmmAll) == "mvwt"
    assert khg.str_to_gmm_flags("mvwt") == khg.GmmUpdateFlags.kGmmAll == khg.str_to_gmm_flags("a")
    with pytest.raises(khg.KhgError):
        khg.str_to_gmm_flags("x")
    for f in (0, 1, 2, 4, 15):
        assert khg.augment_gmm_flags(f) == orc.augment_gmm_flags(f)


def test_accum_diag_gmm_shapes_and_dtypes():
    # python/tests/test_mle_diag_gmm.py:48-90
    g = khg.DiagGmm(nmix=3, dim=4)
    acc = khg.AccumDiagGmm(g, khg.GmmUpdateFlags.kGmmVariances)
    assert acc.flags == 7 and acc.occupancy.dtype == np.float64
    assert acc.mean_accumulator.shape == (3, 4) and acc.variance_accumulator.shape == (3, 4)
    acc = khg.AccumDiagGmm(g, khg.GmmUpdateFlags.kGmmWeights)
    assert acc.mean_accumulator.size == 0 and acc.variance_accumulator.size == 0
    # :92-163 accumulate_for_component / :165-198 accumulate_from_posteriors
    acc = khg.AccumDiagGmm(g, khg.GmmUpdateFlags.kGmmAll)
    x = np.arange(4, dtype=np.float32) + 1
    acc.accumulate_for_component(x, 1, 0.5)
    assert acc.occupancy[1] == 0.5 and np.allclose(acc.mean_accumulator[1], x * 0.5)
    assert np.allclose(acc.variance_accumulator[1], x * x * 0.5)
    post = np.array([0.2, 0.3, 0.5], np.float32)
    before = (acc.occupancy.copy(), acc.mean_accumulator.copy(), acc.variance_accumulator.copy())
    acc.accumulate_from_posteriors(x, post)
    assert np.allclose(acc.occupancy, before[0] + post)
    assert np.allclose(acc.mean_accumulator, before[1] + np.outer(post, x))
    assert np.allclose(acc.variance_accumulator, before[2] + np.outer(post, x * x))
    with pytest.raises(khg.KhgError):
        acc.set_zero(0x10)


def _rand_am(rng, P, G, D):
    am = khg.AmDiagGmm()
    for _ in range(P):
        g = khg.DiagGmm(nmix=G, dim=D)
        w = rng.random(G).astype(np.float32); w /= w.sum()
        g.set_weights(w)
        g.set_invvars((1 / (rng.random((G, D)) + 0.5)).astype(np.float32))
        g.set_means(rng.standard_normal((G, D)).astype(np.float32))
        g.compute_gconsts()
        am.add_pdf(g)
    return am


@pytest.mark.parametrize("flags", ["mvw", "mw", "w", "v"])
def test_m_step_bit_exact_vs_oracle(flags):
    """khg_mle_am_diag_gmm_update (product, C++) == oracle restatement of MleDiagGmmUpdate, bit for bit,
    including low-occupancy removal and variance flooring."""
    rng = np.random.default_rng(5)
    P, G, D = 4, 6, 5
    am = _rand_am(rng, P, G, D)
    accs = khg.AccumAmDiagGmm()
    accs.init(am, khg.GmmUpdateFlags.kGmmAll)
    for p in range(P):
        a = accs._accs[p]
        for _ in range(300):
            x = (rng.standard_normal(D) * 1.2).astype(np.float32)
            post = rng.random(G).astype(np.float32); post /= post.sum()
            a.accumulate_from_posteriors(x, post)
        a.occupancy[1] = 2.0          # -> removed (min_gaussian_occupancy = 10)
        a.variance_accumulator[2] = a.mean_accumulator[2] ** 2 / a.occupancy[2]   # -> zero variance -> floored
    ref = []
    f = int(khg.str_to_gmm_flags(flags))
    for p in range(P):
        g, a = am.get_pdf(p), accs._accs[p]
        ref.append(orc.mle_diag_gmm_update(g.weights, g.means_invvars, g.inv_vars, a.occupancy, a.mean_accumulator,
                                           a.variance_accumulator, acc_flags=a.flags, flags=f))
    objf, count = khg.mle_am_diag_gmm_update(khg.MleDiagGmmOptions(), accs, khg.str_to_gmm_flags(flags), am)
    tot_obj = np.float32(0); tot_cnt = np.float32(0)
    for p in range(P):
        g = am.get_pdf(p)
        assert g.num_gauss == G - 1 == len(ref[p]["weights"])
        for k, arr in (("weights", g.weights), ("gconsts", g.gconsts), ("means_invvars", g.means_invvars), ("inv_vars", g.inv_vars)):
            np.testing.assert_array_equal(arr, ref[p][k], err_msg=f"pdf {p} {k}")
        tot_obj = np.float32(tot_obj + np.float32(ref[p]["obj_change"])); tot_cnt = np.float32(tot_cnt + np.float32(ref[p]["count"]))
    assert objf == tot_obj and count == tot_cnt


def test_transition_mle_update_vs_oracle():
    topo, tree, tm = _tm()
    rng = np.random.default_rng(2)
    stats = tm.init_stats()
    stats[1:] = rng.integers(0, 50, size=tm.num_transition_ids)
    stats[3] = 0; stats[4] = 1        # below mincount -> skipped
    s2i = tm.state2id
    slo = [0] + [tm.self_loop_of(ts) for ts in range(1, tm.num_transition_states + 1)]
    lp, nsl, oi, cnt = orc.transition_mle_update(s2i, slo, stats, tm.log_probs, tm.non_self_loop_log_probs)
    objf, count = tm.mle_update(stats, khg.MleTransitionUpdateConfig())
    np.testing.assert_array_equal(np.asarray(tm.log_probs, np.float32), lp)
    np.testing.assert_array_equal(np.asarray(tm.non_self_loop_log_probs, np.float32), nsl)
    assert objf == oi and count == cnt


def test_scaled_trans_cost_vs_oracle_add_transition_probs():
    topo, tree, tm = _tm()
    for ts, sl in ((1.0, 1.0), (1.0, 0.1), (0.0, 0.5)):
        cost = tm.scaled_trans_cost(ts, sl)
        il = np.arange(tm.num_transition_ids + 1, dtype=np.int32)
        i2s = np.asarray(tm.id2state, np.int32)
        w = orc.add_transition_probs(il, np.zeros(il.shape[0], np.float32), tm.log_probs, tm.non_self_loop_log_probs, i2s,
                                     tm.is_self_loop_array(), ts, sl)
        np.testing.assert_array_equal(cost[1:], w[1:])
    fst = khg.StdVectorFst()
    a, b = fst.add_state(), fst.add_state()
    fst.start = a
    fst.add_arc(a, khg.StdArc(2, 0, 0.5, b)); fst.add_arc(b, khg.StdArc(1, 0, 0.0, b)); fst.add_arc(b, khg.StdArc(0, 7, 1.0, b))
    fst.set_final(b, 0.0)
    khg.add_transition_probs(trans_model=tm, transition_scale=1.0, self_loop_scale=0.1, fst=fst)
    c = tm.scaled_trans_cost(1.0, 0.1)
    assert fst.arcs(a)[0].weight == np.float32(np.float32(0.5) + c[2]) and fst.arcs(b)[0].weight == c[1]
    assert fst.arcs(b)[1].weight == 1.0
    fst.add_arc(b, khg.StdArc(999, 0, 0.0, b))
    with pytest.raises(khg.KhgError):     # csrc/hmm-utils.cc:484-488
        khg.add_transition_probs(trans_model=tm, transition_scale=1.0, self_loop_scale=1.0, fst=fst)


def test_diag_gmm_host_bookkeeping():
    # python/tests/test_diag_gmm.py:14-106 (set/get), :108-167 (remove_component), pickle :819-848
    rng = np.random.default_rng(0)
    nmix, dim = 10, 8
    g = khg.DiagGmm(nmix=nmix, dim=dim)
    w = rng.random(nmix).astype(np.float32); w /= w.sum()
    mean = rng.random((nmix, dim)).astype(np.float32); var = (rng.random((nmix, dim)) + 0.1).astype(np.float32)
    g.set_weights(w); g.set_means(mean); g.set_invvars(1 / var)
    assert np.allclose(g.weights, w) and np.allclose(g.means, mean, atol=1e-6) and np.allclose(g.vars, var, rtol=1e-6)
    assert g.valid_gconsts is False and g.compute_gconsts() == 0 and g.valid_gconsts is True
    expected = np.log(w) - 0.5 * (dim * np.log(2 * np.pi) + np.log(var).sum(1) + (mean ** 2 / var).sum(1))
    assert np.allclose(g.gconsts, expected, rtol=1e-5)
    g.remove_component(3, renorm_weights=True)
    assert g.num_gauss == 9 and abs(g.weights.sum() - 1) < 1e-6 and g.valid_gconsts is False
    g.remove_components([0, 5], renorm_weights=False)
    assert g.num_gauss == 7
    g.compute_gconsts()
    g2 = pickle.loads(pickle.dumps(g, 2))
    assert np.array_equal(g2.weights, g.weights) and np.array_equal(g2.gconsts, g.gconsts) and g2.valid_gconsts
    am = khg.AmDiagGmm(); am.add_pdf(g); am.add_pdf(g2)
    am2 = pickle.loads(pickle.dumps(am, 2))
    assert am2.num_pdfs == 2 and am2.num_gauss == 14 and np.array_equal(am2.get_pdf(1).means_invvars, g.means_invvars)
    h = []
    g3 = khg.DiagGmm(gmm=g); g3.split(9, 0.01, history=h, randn=lambda d: np.ones(d, np.float32))
    assert g3.num_gauss == 9 and len(h) == 2 and abs(g3.weights.sum() - g.weights.sum()) < 1e-6   # :169-237


def test_get_split_targets_power_rule():
    # csrc/model-common.cc:29-70
    t = khg.get_split_targets(np.array([1000.0, 10.0, 0.0], np.float32), 10, 0.2, 20.0)
    assert sum(t) <= 10 and t[0] > t[1] >= 1 and t[2] == 1
    assert khg.get_split_targets(np.array([100.0, 100.0], np.float32), 6, 0.2, 20.0) == [3, 3]


def test_shard_utterances_balances_frames():
    from kaldi_hmm_gmm_amd.dist import shard_utterances
    T = np.random.default_rng(0).integers(100, 500, size=1000)
    shards = shard_utterances(T, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(1000))
    loads = [T[s].sum() for s in shards]
    assert max(loads) - min(loads) <= T.max()


def test_diag_gmm_generate_and_interpolate():
    """csrc/diag-gmm.cc:410-446 (Generate) and :460-484 (Interpolate) on the host object (no GPU work)."""
    rng = np.random.default_rng(3)
    g = khg.DiagGmm(nmix=3, dim=4)
    g.set_weights(np.array([.2, .3, .5], np.float32)); g.set_means(rng.standard_normal((3, 4)).astype(np.float32))
    g.set_invvars(np.full((3, 4), 2.0, np.float32)); g.compute_gconsts()
    h = khg.DiagGmm(nmix=3, dim=4)
    h.set_weights(np.array([.5, .3, .2], np.float32)); h.set_means(np.zeros((3, 4), np.float32))
    h.set_invvars(np.ones((3, 4), np.float32)); h.compute_gconsts()
    # a draw of exactly 0 picks component 0 and returns its mean
    x = g.generate(randn=lambda shape: np.zeros(shape, np.float32))
    assert x.dtype == np.float32 and np.allclose(x, g.means[0], atol=1e-6)
    # a large positive draw runs off the end -> last component; unit deviates add one standard deviation
    x = g.generate(randn=lambda shape: np.ones(shape, np.float32) * (5.0 if np.prod(shape) == 1 else 1.0))
    assert np.allclose(x, g.means[2] + np.sqrt(0.5), atol=1e-5)
    m0 = g.means.copy()
    g.interpolate(0.25, h)
    assert np.allclose(g.means, 0.75 * m0, atol=1e-6) and np.allclose(g.vars, 0.75 * 0.5 + 0.25)
    assert np.allclose(g.weights, [0.275, 0.3, 0.425], atol=1e-6) and g.valid_gconsts
    w0 = g.weights.copy()
    g.interpolate(0.5, h, flags=khg.GmmUpdateFlags.kGmmMeans)
    assert np.allclose(g.weights, w0) and np.allclose(g.means, 0.375 * m0, atol=1e-6)
    with pytest.raises(khg.KhgError):
        g.interpolate(0.5, khg.DiagGmm(nmix=2, dim=4))


def test_accum_diag_gmm_smoothing():
    # csrc/mle-diag-gmm.cc:192-241 (SmoothStats / SmoothWithAccum / SmoothWithModel)
    rng = np.random.default_rng(11)
    am = _rand_am(rng, 1, 3, 4)
    g = am.get_pdf(0)
    acc = khg.AccumDiagGmm(g, khg.GmmUpdateFlags.kGmmAll)
    for _ in range(20):
        acc.accumulate_from_posteriors(rng.standard_normal(4).astype(np.float32), rng.dirichlet(np.ones(3)).astype(np.float32))
    occ, m, v = acc.occupancy.copy(), acc.mean_accumulator.copy(), acc.variance_accumulator.copy()
    a = acc.copy()
    a.smooth_stats(2.0)
    assert np.allclose(a.occupancy, occ + 2.0)
    # per-Gaussian means (mean_acc / occ) are unchanged by adding virtual counts of the acc's own stats
    assert np.allclose(a.mean_accumulator / a.occupancy[:, None], m / occ[:, None])
    assert np.allclose(a.variance_accumulator / a.occupancy[:, None], v / occ[:, None])
    src = acc.copy()
    src.scale(3.0, 7)
    src.occupancy[1] = 0.0            # a source Gaussian without data is skipped
    b = acc.copy()
    b.smooth_with_accum(4.0, src)
    assert np.allclose(b.occupancy, occ + np.array([4.0, 0.0, 4.0]))
    assert np.allclose(b.mean_accumulator[0], m[0] + src.mean_accumulator[0] * 4.0 / src.occupancy[0])
    assert np.array_equal(b.mean_accumulator[1], m[1]) and np.array_equal(b.variance_accumulator[1], v[1])
    c = acc.copy()
    c.smooth_with_model(1.5, g)
    mu, var = g.means.astype(np.float64), g.vars.astype(np.float64)
    assert np.allclose(c.occupancy, occ + 1.5)
    assert np.allclose(c.mean_accumulator, m + 1.5 * mu)
    assert np.allclose(c.variance_accumulator, v + 1.5 * (var + mu * mu))
    with pytest.raises(khg.KhgError):
        c.smooth_with_model(1.0, khg.DiagGmm(nmix=2, dim=4))


def _set_gmm(w, mean, var):
    g = khg.DiagGmm(nmix=len(w), dim=mean.shape[1])
    g.set_weights(np.asarray(w, np.float32))
    g.set_invvars((1 / var).astype(np.float32))        # python/tests/test_diag_gmm.py:248-250 order: means, then invvars
    g.set_means(mean.astype(np.float32))
    g.compute_gconsts()
    return g


def test_diag_gmm_merge_reference_known_answers():
    """The reference's own checks of DiagGmm::Merge (python/tests/test_diag_gmm.py:239-263 and :265-299), against the product
    (khg_diag_gmm_merge through DiagGmm.merge) AND the oracle restatement."""
    rng = np.random.default_rng(7)
    nmix, dim = 7, 6
    w = rng.random(nmix).astype(np.float32); w /= w.sum()
    mean = rng.random((nmix, dim)).astype(np.float32)
    var = (rng.random((nmix, dim)) + 0.05).astype(np.float32)
    g = _set_gmm(w, mean, var)
    o = orc.diag_gmm_merge(g.weights, g.means_invvars, g.inv_vars, 1)
    history = g.merge(target_components=1)
    assert history == [] and o["history"] == []                         # :254
    assert abs(g.weights[0] - 1) < 1e-6 and g.num_gauss == 1            # :255-256
    exp_mean = (w.astype(np.float64)[None] @ mean.astype(np.float64))
    exp_var = w.astype(np.float64)[None] @ (var.astype(np.float64) + mean.astype(np.float64) ** 2) - exp_mean ** 2
    np.testing.assert_allclose(g.means, exp_mean, rtol=1e-5, atol=1e-8)   # :258-263 (torch.allclose defaults)
    np.testing.assert_allclose(g.vars, exp_var, rtol=1e-4, atol=1e-6)
    for k, arr in (("weights", g.weights), ("gconsts", g.gconsts), ("means_invvars", g.means_invvars), ("inv_vars", g.inv_vars)):
        np.testing.assert_array_equal(arr, o[k], err_msg=k)
    # case 2 (:265-299): components 2 and 0 are the closest pair; "2 comes first before 0 in history since we are building a
    # lower triangular matrix in C++"
    w = rng.random(4).astype(np.float32); w /= w.sum()
    mean = np.array([[2, 2], [-10, -10], [1, 1], [-100, 100]], np.float32)
    var = (rng.random((4, 2)) + 0.05).astype(np.float32)
    g = _set_gmm(w, mean, var)
    o = orc.diag_gmm_merge(g.weights, g.means_invvars, g.inv_vars, 3)
    assert g.merge(target_components=3) == [2, 0] == o["history"] and g.num_gauss == 3
    with pytest.raises(khg.KhgError):
        g.merge(target_components=5)                                    # diag-gmm.cc:558-561
    assert g.merge(target_components=3) == []                           # :563-567 nothing to do


@pytest.mark.parametrize("G,D,target", [(9, 5, 4), (16, 13, 1), (16, 13, 15), (33, 40, 7), (5, 1, 2), (64, 40, 48)])
def test_diag_gmm_merge_bit_exact_vs_oracle(G, D, target):
    rng = np.random.default_rng(G * 100 + D)
    w = rng.random(G).astype(np.float32); w /= w.sum()
    mean = (2.0 * rng.standard_normal((G, D))).astype(np.float32)
    mean[G // 2] = mean[0] + 0.01                                       # a nearly identical pair
    var = (rng.random((G, D)) * 1.5 + 0.3).astype(np.float32)
    g = _set_gmm(w, mean, var)
    o = orc.diag_gmm_merge(g.weights, g.means_invvars, g.inv_vars, target)
    hist = g.merge(target)
    assert hist == o["history"] and len(hist) == (2 * (G - target) if target > 1 else 0) and g.num_gauss == target
    for k, arr in (("weights", g.weights), ("gconsts", g.gconsts), ("means_invvars", g.means_invvars), ("inv_vars", g.inv_vars)):
        np.testing.assert_array_equal(arr, o[k], err_msg=k)
    assert abs(float(g.weights.sum()) - 1.0) < 1e-5
    if target > 1:
        assert all(hist[k] > hist[k + 1] for k in range(0, len(hist), 2))    # (max_i, max_j) pairs come from the lower triangle: j < i


def test_am_diag_gmm_merge_by_count_and_gmm_est_mixdown():
    """AmDiagGmm::MergeByCount (csrc/am-diag-gmm.cc:91-108): per-pdf targets from GetSplitTargets, never below 1."""
    rng = np.random.default_rng(3)
    am = _rand_am(rng, 5, 6, 4)
    occs = np.array([500.0, 20.0, 0.0, 3000.0, 100.0], np.float32)
    from kaldi_hmm_gmm_amd.mle import get_split_targets
    targets = get_split_targets(occs, 14, 0.2, 20.0)
    before = [am.get_pdf(p).num_gauss for p in range(5)]
    am.merge_by_count(state_occs=occs, target_components=14, power=0.2, min_count=20.0)
    after = [am.get_pdf(p).num_gauss for p in range(5)]
    assert after == [min(b, max(t, 1)) for b, t in zip(before, targets)] and sum(after) < sum(before)
    for p in range(5):
        assert abs(float(am.get_pdf(p).weights.sum()) - 1.0) < 1e-5 and np.isfinite(am.get_pdf(p).gconsts).all()


def test_m_step_variance_floor_vector_bit_exact_vs_oracle():
    """MleDiagGmmOptions.variance_floor_vector (csrc/mle-diag-gmm.h:26-28, mle-diag-gmm.cc:311-322): per-dimension floors
    replace min_variance; host C++ M-step == oracle bit for bit, floored counters included."""
    rng = np.random.default_rng(15)
    P, G, D = 3, 5, 6
    am = _rand_am(rng, P, G, D)
    accs = khg.AccumAmDiagGmm()
    accs.init(am, khg.GmmUpdateFlags.kGmmAll)
    for p in range(P):
        a = accs._accs[p]
        for _ in range(400):
            x = (rng.standard_normal(D) * np.array([0.05, 1, 1, 0.2, 1, 3])).astype(np.float32)
            post = rng.random(G).astype(np.float32); post /= post.sum()
            a.accumulate_from_posteriors(x, post)
    floor = np.array([0.5, 1e-4, 1e-4, 0.2, 1e-4, 1e-4], np.float64)    # dims 0 and 3 get floored, min_variance is ignored
    opts = khg.MleDiagGmmOptions(min_variance=100.0, variance_floor_vector=floor)
    ref = [orc.mle_diag_gmm_update(am.get_pdf(p).weights, am.get_pdf(p).means_invvars, am.get_pdf(p).inv_vars, accs._accs[p].occupancy,
                                   accs._accs[p].mean_accumulator, accs._accs[p].variance_accumulator, acc_flags=accs._accs[p].flags,
                                   flags=7, min_variance=100.0, variance_floor_vector=floor) for p in range(P)]
    khg.mle_am_diag_gmm_update(opts, accs, khg.str_to_gmm_flags("mvw"), am)
    for p in range(P):
        g = am.get_pdf(p)
        assert ref[p]["floored_elems"] == 2 * G and ref[p]["floored_gauss"] == G
        for k, arr in (("weights", g.weights), ("gconsts", g.gconsts), ("means_invvars", g.means_invvars), ("inv_vars", g.inv_vars)):
            np.testing.assert_array_equal(arr, ref[p][k], err_msg=f"pdf {p} {k}")
        np.testing.assert_allclose(g.vars[:, 0], 0.5, rtol=1e-6)
        np.testing.assert_allclose(g.vars[:, 3], 0.2, rtol=1e-6)
        assert (g.vars[:, 5] > 1.0).all()


def test_gmm_boost_silence_returns_a_boosted_copy():
    """scripts/gmm_boost_silence.py:10-45 returns a NEW AmDiagGmm and leaves its argument untouched; the reference's recipe
    does `am = gmm_boost_silence(am_gmm=am, ...)` (egs/yesno/train.py:158)."""
    topo, tree, tm = _tm()
    rng = np.random.default_rng(9)
    am = _rand_am(rng, tm.num_pdfs, 3, 4)
    w0 = [am.get_pdf(p).weights.copy() for p in range(am.num_pdfs)]
    gc0 = [am.get_pdf(p).gconsts.copy() for p in range(am.num_pdfs)]
    phones = [3, 1]
    out = khg.gmm_boost_silence(am_gmm=am, transition_model=tm, silence_phones=phones, boost=1.5)
    assert out is not am and isinstance(out, khg.AmDiagGmm) and phones == [1, 3]      # sorted in place, like the reference
    _, pdfs = khg.get_pdfs_for_phones(tm, [1, 3])
    for p in range(am.num_pdfs):
        assert np.array_equal(am.get_pdf(p).weights, w0[p]) and np.array_equal(am.get_pdf(p).gconsts, gc0[p])
        if p in pdfs:
            np.testing.assert_array_equal(out.get_pdf(p).weights, w0[p] * np.float32(1.5))
            np.testing.assert_allclose(out.get_pdf(p).gconsts, gc0[p] + np.log(1.5), rtol=1e-6, atol=1e-5)
        else:
            assert np.array_equal(out.get_pdf(p).weights, w0[p]) and np.array_equal(out.get_pdf(p).gconsts, gc0[p])


def test_careful_graph_c_abi_vs_oracle_and_python_fst():
    """khg_careful_graph (host C-ABI: csrc/decoder-wrappers.cc:111-140 + OpenFst Concat) == the oracle's restatement ==
    the package's StdVectorFst form, on random graphs with several final states, and an empty graph."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphs import random_graph
    from kaldi_hmm_gmm_amd.fst import StdArc, StdVectorFst, modify_graph_for_careful_alignment
    from kaldi_hmm_gmm_amd._lib import lib, ptr, check
    rng = np.random.default_rng(21)
    for trial in range(12):
        g = random_graph(rng, 40, n_main=int(rng.integers(1, 12)), p_branch=0.4, p_eps=0.3)
        S = len(g["final"]); A = len(g["ilabel"])
        if trial % 3 == 0 and S > 2:
            g["final"][S // 2] = np.float32(0.25)              # a second final state
        start = int(g["start"])
        o = orc.careful_graph(orc.OGraph(start, g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"]))
        ao = np.zeros(2 * S + 2, np.int64); il = np.zeros(2 * A + S + 1, np.int32); ol = np.zeros_like(il)
        w = np.zeros(2 * A + S + 1, np.float32); ns = np.zeros_like(il); fin = np.zeros(2 * S + 1, np.float32)
        nS, st = C.c_int32(), C.c_int32()
        a64 = np.ascontiguousarray(g["arc_off"], np.int64)
        check(lib.khg_careful_graph(S, start, ptr(a64, C.c_int64), ptr(g["ilabel"], C.c_int32), ptr(g["olabel"], C.c_int32),
                                    ptr(g["weight"], C.c_float), ptr(g["nextstate"], C.c_int32), ptr(g["final"], C.c_float),
                                    C.byref(nS), C.byref(st), ptr(ao, C.c_int64), ptr(il, C.c_int32), ptr(ol, C.c_int32),
                                    ptr(w, C.c_float), ptr(ns, C.c_int32), ptr(fin, C.c_float)))
        n = int(ao[nS.value])
        assert nS.value == 2 * S + 1 == o.final.shape[0] and st.value == start == o.c.start
        assert n == 2 * A + 1 + int(np.isfinite(g["final"]).sum()) == o.ilabel.shape[0]
        assert np.array_equal(ao[: nS.value + 1], o.arc_off) and np.array_equal(fin, o.final)
        for got, want in ((il[:n], o.ilabel), (ol[:n], o.olabel), (w[:n], o.weight), (ns[:n], o.nextstate)):
            assert np.array_equal(got, want)
        # the package's graph class (what AlignConfig.careful uses above the C-ABI)
        f = StdVectorFst()
        for _ in range(S):
            f.add_state()
        f.start = start
        for s in range(S):
            for a in range(int(g["arc_off"][s]), int(g["arc_off"][s + 1])):
                f.add_arc(s, StdArc(int(g["ilabel"][a]), int(g["olabel"][a]), float(g["weight"][a]), int(g["nextstate"][a])))
            if np.isfinite(g["final"][s]):
                f.set_final(s, float(g["final"][s]))
        modify_graph_for_careful_alignment(f)
        assert f.num_states == nS.value
        flat = [(a.ilabel, a.olabel, np.float32(a.weight), a.nextstate) for s in range(f.num_states) for a in f._arcs[s]]
        assert flat == list(zip(il[:n].tolist(), ol[:n].tolist(), w[:n], ns[:n].tolist()))
    nS, st = C.c_int32(7), C.c_int32(7)
    ao = np.ones(2, np.int64)
    check(lib.khg_careful_graph(0, -1, None, None, None, None, None, None, C.byref(nS), C.byref(st), ptr(ao, C.c_int64), None, None,
                                None, None, None))
    assert nS.value == 0 and st.value == -1 and ao[0] == 0


def test_pybind11_module_is_the_host_surface():
    """The pybind11 module `_kaldi_hmm_gmm_amd` (csrc/khg_pybind.cpp; the reference's boundary is `_kaldi_hmm_gmm`,
    python/csrc/kaldi-hmm-gmm.cc:35-69) is built, is what kaldi_hmm_gmm_amd's device classes are, raises KhgError like
    the reference's RuntimeError, and its host functions agree with the ctypes route to the same C-ABI."""
    from kaldi_hmm_gmm_amd import device
    ext = pytest.importorskip("kaldi_hmm_gmm_amd._kaldi_hmm_gmm_amd")
    assert device.BINDING == "pybind11"
    assert khg.Context is ext.Context and khg.UtteranceSet is ext.UtteranceSet and khg.DeviceModel is ext.DeviceModel
    assert issubclass(khg.DeviceAccs, ext.DeviceAccs) and khg.Comm is ext.Comm
    for cls, methods in ((ext.Context, "sync set_timing timings set_k1_form set_option get_option close"),
                         (ext.DeviceModel, "set_weights mle_update scale_weights download close"),
                         (ext.DeviceTransitions, "set_trans_cost close"),
                         (ext.UtteranceSet, "set_pdf_list pdf_lists pdf_first_frames loglikes loglikes_layout download_loglikes "
                                            "upload_loglikes align upload_ali download_ali acc_stats close"),
                         (ext.DeviceAccs, "zero device_ptr allreduce split relayout download_range download_occ download_trans download "
                                          "upload close"),
                         (ext.Comm, "unique_id close")):
        for name in methods.split():
            assert hasattr(cls, name), (cls, name)
    assert ext.version() == 100 and issubclass(khg.KhgError, RuntimeError)
    # the reference-named host classes ARE the extension's C++ classes (csrc/khg_host_{gmm,hmm,align}.cpp), not Python mirrors
    for name in ("DiagGmm", "AmDiagGmm", "AccumDiagGmm", "AccumAmDiagGmm", "MleDiagGmmOptions", "HmmState", "HmmTopology",
                 "TransitionModelTuple", "TransitionModel", "MleTransitionUpdateConfig", "AlignConfig", "FasterDecoderOptions",
                 "DecodableAmDiagGmmUnmapped", "DecodableAmDiagGmmScaled", "DecodableInterface", "TransitionInformation", "MapDiagGmmOptions",
                 "FasterDecoder", "LatticeWeight", "LatticeArc", "LinearLattice", "StdVectorFst", "StdArc"):
        assert getattr(khg, name) is getattr(ext, name), name
    assert khg.align_utterance_wrapper is ext.align_utterance_wrapper and khg.align_batch is ext.align_batch
    assert khg.add_transition_probs is ext.add_transition_probs and khg.modify_graph_for_careful_alignment is ext.modify_graph_for_careful_alignment
    assert khg.get_pdfs_for_phones is ext.get_pdfs_for_phones and khg.ml_objective is ext.ml_objective
    for cls, methods in ((ext.DiagGmm, "resize set_weights set_means set_invvars compute_gconsts log_likelihood log_likelihoods component_posteriors "
                                       "split merge perturb generate interpolate remove_component"),
                         (ext.AmDiagGmm, "init add_pdf get_pdf split_by_count merge_by_count compute_gconsts flat set_flat"),
                         (ext.AccumDiagGmm, "resize accumulate_from_diag accumulate_from_posteriors accumulate_for_component add smooth_stats"),
                         (ext.AccumAmDiagGmm, "init accumulate_for_gmm accumulate_for_gmm_two_feats accumulate_from_posteriors add scale get_acc"),
                         (ext.HmmTopology, "read check topology_for_phone num_pdf_classes min_length"),
                         (ext.TransitionModel, "transition_id_to_pdf is_self_loop mle_update scaled_trans_cost tuple_to_transition_state "
                                               "pair_to_transition_id get_transition_log_prob_ignoring_self_loops")):
        for name in methods.split():
            assert hasattr(cls, name), (cls, name)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(khg.KhgError, match="no HIP device"):
            ext.Context(0)
    rng = np.random.default_rng(4)
    am = _rand_am(rng, 3, 4, 5)
    go, gc, w, miv, iv = am.flat()
    gc2, bad = ext.compute_gconsts(go, w, iv, miv)
    assert bad == 0 and np.array_equal(gc2, gc)
    g = am.get_pdf(1)
    r = ext.diag_gmm_merge(g.weights, g.means_invvars, g.inv_vars, 2)
    o = orc.diag_gmm_merge(g.weights, g.means_invvars, g.inv_vars, 2)
    assert r["num_gauss"] == 2 and r["history"] == o["history"]
    for k in ("weights", "gconsts", "means_invvars", "inv_vars"):
        np.testing.assert_array_equal(r[k][:2], o[k])
    topo, tree, tm = _tm()
    c = ext.scaled_trans_cost(np.asarray(tm.log_probs, np.float32), np.asarray(tm.non_self_loop_log_probs, np.float32),
                              np.asarray(tm.id2state, np.int32), np.asarray([0] + [int(tm.is_self_loop(t)) for t in range(1, tm.num_transition_ids + 1)], np.uint8), 1.0, 0.1)
    np.testing.assert_array_equal(c, tm.scaled_trans_cost(1.0, 0.1))


def test_transition_mle_update_shared_for_pdfs():
    """MleTransitionUpdateConfig(share_for_pdfs=True) (transition-model.cc:531-655): transition-states that share a pdf get ONE
    set of probabilities from their pooled counts.  Checked against the oracle's restatement bit for bit, and through two
    properties: with one transition-state per pdf it is the ordinary update (to float rounding of the objective), and with
    a tree that ties two phones the tied states end up with identical probabilities."""
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd.training_graph import generate_hmm_topo

    topo = generate_hmm_topo(non_sil_phones=[2, 3, 4], sil_phone=1)
    rng = np.random.default_rng(5)

    def model(shared_phones=None):
        p2n = topo.get_phone_to_num_pdf_classes()
        tree = khg.monophone_context_dependency_shared(shared_phones, p2n) if shared_phones else khg.monophone_context_dependency(topo.phones, p2n)
        return khg.TransitionModel(tree, topo)

    # (1) one transition-state per pdf: shared == ordinary
    tm_a, tm_b = model(), model()
    stats = rng.uniform(0.0, 40.0, tm_a.num_transition_ids + 1)
    stats[0] = 0.0
    stats[tm_a.pair_to_transition_id(3, 0): tm_a.pair_to_transition_id(3, 0) + 2] = [1.0, 0.5]      # below mincount: skipped
    oa, ca = tm_a.mle_update(stats, khg.MleTransitionUpdateConfig())
    ob, cb = tm_b.mle_update(stats, khg.MleTransitionUpdateConfig(share_for_pdfs=True))
    np.testing.assert_allclose(tm_b.log_probs, tm_a.log_probs, rtol=0, atol=1e-7)
    np.testing.assert_allclose(tm_b.non_self_loop_log_probs, tm_a.non_self_loop_log_probs, rtol=0, atol=1e-6)
    assert cb == pytest.approx(ca, rel=1e-6) and ob == pytest.approx(oa, rel=1e-4, abs=1e-3)
    # against the oracle, bit for bit
    tm_c = model()
    s2i = np.asarray(tm_c._state2id, np.int64)
    fwd = np.asarray([0] + [t.forward_pdf for t in tm_c._tuples], np.int64)
    want_lp, want_o, want_c = orc.transition_mle_update_shared(s2i, fwd, stats, tm_c.log_probs)
    oc, cc = tm_c.mle_update(stats, khg.MleTransitionUpdateConfig(share_for_pdfs=True))
    assert np.array_equal(tm_c.log_probs, want_lp) and oc == want_o and cc == want_c
    # (2) phones 2 and 3 tied by the tree: their states share pdfs -> identical probabilities from pooled counts
    tm_t = model(shared_phones=[[1], [2, 3], [4]])
    s2i = np.asarray(tm_t._state2id, np.int64)
    fwd = np.asarray([0] + [t.forward_pdf for t in tm_t._tuples], np.int64)
    stats = rng.uniform(5.0, 40.0, tm_t.num_transition_ids + 1)
    want_lp, want_o, want_c = orc.transition_mle_update_shared(s2i, fwd, stats, tm_t.log_probs)
    ot, ct = tm_t.mle_update(stats, khg.MleTransitionUpdateConfig(share_for_pdfs=True))
    assert np.array_equal(tm_t.log_probs, want_lp) and ot == want_o and ct == want_c
    tied = {}
    for ts in range(1, tm_t.num_transition_states + 1):
        tied.setdefault(tm_t._tuples[ts - 1].forward_pdf, []).append(ts)
    assert any(len(v) > 1 for v in tied.values())
    for v in tied.values():
        for ts in v[1:]:
            n = s2i[ts + 1] - s2i[ts]
            assert np.array_equal(tm_t.log_probs[s2i[ts]: s2i[ts] + n], tm_t.log_probs[s2i[v[0]]: s2i[v[0]] + n])
        # pooled counts: exp(log_prob) = sum of counts over the tied states / total (none floored here)
        n = int(s2i[v[0] + 1] - s2i[v[0]])
        if n > 1:
            pooled = sum(stats[s2i[ts]: s2i[ts] + n] for ts in v)
            np.testing.assert_allclose(np.exp(tm_t.log_probs[s2i[v[0]]: s2i[v[0]] + n]), pooled / pooled.sum(), rtol=1e-6)


def test_map_update_vs_oracle_and_reference_option_tests():
    """MapDiagGmmOptions as the reference's own tests pin it (python/tests/test_mle_diag_gmm.py:34-46), and map_diag_gmm_update /
    map_am_diag_gmm_update (csrc/mle-diag-gmm.cc:392-477, csrc/mle-am-diag-gmm.cc:204-227; python/csrc/mle-diag-gmm.cc:136-151) against
    the oracle's numpy restatement: parameters bit for bit for every flag combination, zero-occupancy components left at their
    prior, the objective change positive on statistics drawn away from the model."""
    opts = khg.MapDiagGmmOptions()
    assert abs(opts.mean_tau - 10) < 1e-5 and abs(opts.variance_tau - 50) < 1e-5 and abs(opts.weight_tau - 10) < 1e-5
    opts = khg.MapDiagGmmOptions(mean_tau=1, variance_tau=2, weight_tau=3)
    assert abs(opts.mean_tau - 1) < 1e-5 and abs(opts.variance_tau - 2) < 1e-5 and abs(opts.weight_tau - 3) < 1e-5
    assert str(opts) == "MapDiagGmmOptions(mean_tau=1, variance_tau=2, weight_tau=3)"
    rng = np.random.default_rng(12)
    am = _rand_am(rng, 4, 6, 7)
    accs = khg.AccumAmDiagGmm(); accs.init(am, khg.GmmUpdateFlags.kGmmAll)
    for p in range(am.num_pdfs):
        g, a = am.get_pdf(p), accs._accs[p]
        occ = rng.uniform(0.0, 40.0, g.num_gauss)
        occ[0] = 0.0                                    # no data: the component keeps its mean and variance
        mu = g.means.astype(np.float64) + rng.normal(0, 0.5, g.means.shape)
        var = g.vars.astype(np.float64) * rng.uniform(0.5, 2.0, g.vars.shape)
        a.occupancy[:] = occ
        a.mean_accumulator[:] = occ[:, None] * mu
        a.variance_accumulator[:] = occ[:, None] * (var + mu * mu)
    cfg = khg.MapDiagGmmOptions(mean_tau=4.0, variance_tau=20.0, weight_tau=8.0)
    for flags in ("mvw", "mw", "vw", "w", "mv", "m"):
        f = khg.str_to_gmm_flags(flags)
        am2 = khg.AmDiagGmm(); am2.copy_from_am_diag_gmm(am)
        tot_obj, tot_cnt = np.float32(0), np.float32(0)
        for p in range(am.num_pdfs):
            g = khg.DiagGmm(gmm=am.get_pdf(p))
            a = accs.get_acc(p)
            want = orc.map_diag_gmm_update(g.weights, g.means_invvars, g.inv_vars, a.occupancy, a.mean_accumulator, a.variance_accumulator,
                                           flags=int(f), mean_tau=4.0, variance_tau=20.0, weight_tau=8.0)
            before = (g.weights, g.means_invvars, g.inv_vars)
            obj, cnt = khg.map_diag_gmm_update(cfg, a, f, g)
            np.testing.assert_array_equal(g.weights, want["weights"]); np.testing.assert_array_equal(g.inv_vars, want["inv_vars"])
            np.testing.assert_array_equal(g.means_invvars, want["means_invvars"]); np.testing.assert_array_equal(g.gconsts, want["gconsts"])
            assert cnt == want["count"] and g.valid_gconsts
            if not int(f) & 4:
                np.testing.assert_array_equal(g.weights, before[0])
            if not int(f) & 2:
                np.testing.assert_array_equal(g.inv_vars, before[2])
            if int(f) & 1 and not int(f) & 2:
                np.testing.assert_array_equal(g.means[0], am.get_pdf(p).means[0])      # occ = 0: the prior mean
            if flags == "mvw":
                assert obj > 0
            tot_obj = np.float32(tot_obj + np.float32(obj)); tot_cnt = np.float32(tot_cnt + np.float32(cnt))
        obj_am, cnt_am = khg.map_am_diag_gmm_update(cfg, accs, f, am2)
        assert obj_am == float(tot_obj) and cnt_am == float(tot_cnt)
    with pytest.raises(khg.KhgError, match="Flags in argument do not match"):
        a = khg.AccumDiagGmm(am.get_pdf(0), khg.GmmUpdateFlags.kGmmWeights)
        khg.map_diag_gmm_update(cfg, a, khg.GmmUpdateFlags.kGmmAll & 7, khg.DiagGmm(gmm=am.get_pdf(0)))


def test_mutation_versions_follow_every_way_of_changing_a_model():
    """AmDiagGmm caches its device model by a mutation version (khg_host_gmm.hpp): every mutator of a DiagGmm -- reached directly or
    through the reference-returning get_pdf(i) -- and every change of the list of pdfs must move it; readers must not."""
    import pickle

    rng = np.random.default_rng(3)
    g = khg.DiagGmm(nmix=3, dim=4)
    g.set_weights(np.full(3, 1 / 3, np.float32)); g.set_means(rng.standard_normal((3, 4)).astype(np.float32)); g.set_invvars(np.ones((3, 4), np.float32))
    g.compute_gconsts()
    am = khg.AmDiagGmm()
    am.init(g, 4)
    seen = [am._version]

    def moved(what):
        v = am._version
        assert v > seen[-1], what
        seen.append(v)

    def same(what):
        assert am._version == seen[-1], what

    _ = am.get_pdf(1).weights, am.get_pdf(1).means, am.num_gauss, am.flat(), pickle.dumps(am), am.get_gaussian_mean(0, 1), am._pdfs
    same("readers")
    am.get_pdf(2).set_weights(np.asarray([0.5, 0.25, 0.25], np.float32)); moved("set_weights through get_pdf")
    am.get_pdf(2).compute_gconsts(); moved("compute_gconsts")
    am.get_pdf(0).set_component_mean(1, np.zeros(4, np.float32)); moved("set_component_mean")
    am.get_pdf(0).set_component_inv_var(1, np.full(4, 2.0, np.float32)); moved("set_component_inv_var")
    am.get_pdf(0).set_component_weight(0, 0.4); moved("set_component_weight")
    am.get_pdf(0).set_invvars_and_means(np.ones((3, 4), np.float32), np.zeros((3, 4), np.float32)); moved("set_invvars_and_means")
    am.get_pdf(0).compute_gconsts(); moved("compute_gconsts again")
    am.set_gaussian_mean(1, 0, np.ones(4, np.float32)); moved("set_gaussian_mean")
    am.get_pdf(1).compute_gconsts(); moved("gconsts")
    am.get_pdf(3).perturb(0.01, randn=lambda shape: np.zeros(shape, np.float32)); moved("perturb")
    am.get_pdf(3).split(4, 0.01, randn=lambda n: np.zeros(n, np.float32)); moved("split")
    am.get_pdf(3).merge(2); moved("merge")
    am.get_pdf(3).remove_component(0, True); moved("remove_component")
    am.get_pdf(3).compute_gconsts(); moved("gconsts")
    am.get_pdf(1).interpolate(0.5, am.get_pdf(0)); moved("interpolate")
    am.get_pdf(1).copy_from_diag_gmm(am.get_pdf(0)); moved("copy_from_diag_gmm")
    am.get_pdf(1).resize(2, 4); moved("resize")
    am.get_pdf(1).copy_from_diag_gmm(am.get_pdf(0)); moved("copy back")
    am.add_pdf(g); moved("add_pdf")
    am.split_by_count(np.full(5, 100.0, np.float32), 20, 0.01, 0.2, 1.0, randn=lambda n: np.zeros(n, np.float32)); moved("split_by_count")
    am.merge_by_count(np.full(5, 100.0, np.float32), 8, 0.2, 1.0); moved("merge_by_count")
    go, gc, w, miv, iv = am.flat()
    am.set_flat(go, w, gc, miv, iv); moved("set_flat")
    other = khg.AmDiagGmm(); other.copy_from_am_diag_gmm(am)
    same("being copied from")
    am.copy_from_am_diag_gmm(other); moved("copy_from_am_diag_gmm")
    am._pdfs = other._pdfs; moved("replacing the list of pdfs")
    # a pdf shared with another model: a mutation through either is seen by both
    v_other = other._version
    am.get_pdf(0).set_weights(am.get_pdf(0).weights); moved("shared pdf")
    assert other._version > v_other
