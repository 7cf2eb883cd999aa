"""GPU tests through the reference's own names (they read like python/tests/test_diag_gmm.py,
test_mle_diag_gmm.py and scripts/test_gmm_*.py) plus the decoder corner cases: forced fallback to the
order-faithful kernel, epsilon arcs + words, careful mode, exact ties, error / retry statuses."""
import math

import numpy as np
import pytest

from graphs import concat, hub_graph, permute_states, random_graph
from helpers import build, oracle_graph, utt_feats
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def khg(ctx):
    import kaldi_hmm_gmm_amd as k
    from kaldi_hmm_gmm_amd import _gpu

    _gpu.set_default_context(ctx)
    return k


def _gmm(khg, rng, nmix, dim):
    g = khg.DiagGmm(nmix=nmix, dim=dim)
    w = rng.random(nmix).astype(np.float32); w /= w.sum()
    mean = rng.random((nmix, dim)).astype(np.float32)
    var = (rng.random((nmix, dim)) * 0.9 + 0.1).astype(np.float32)
    g.set_weights(w); g.set_means(mean); g.set_invvars(1 / var); g.compute_gconsts()
    return g, w, mean, var


def _density(w, mean, var, x):
    w, mean, var, x = (a.astype(np.float64) for a in (w, mean, var, x))
    e = np.exp(((x - mean) ** 2 / (-2 * var)).sum(1)) / np.sqrt((var * math.pi * 2).prod(1))
    return np.log(w * e)


def test_log_likes_like_reference_test(khg):
    # python/tests/test_diag_gmm.py:327-403,528-576
    rng = np.random.default_rng(20230414)
    g, w, mean, var = _gmm(khg, rng, 10, 8)
    x = rng.random(8).astype(np.float32)
    per = _density(w, mean, var, x)
    assert abs(g.log_likelihood(x) - np.log(np.exp(per).sum())) < 1e-4
    assert np.allclose(g.log_likelihoods(x), per, atol=1e-4)
    X = rng.random((3, 8)).astype(np.float32)
    got = g.log_likelihoods_matrix(X)
    assert got.shape == (3, 10)
    for i in range(3):
        assert np.allclose(got[i], _density(w, mean, var, X[i]), atol=1e-4)
    idx = [0, 1, 3, 8, 7, 8, 3, 2]
    assert np.allclose(g.log_likelihoods_preselect(x, idx), per[idx], atol=1e-4)
    ll, post = g.component_posteriors(x)
    sm = np.exp(per - per.max()); sm /= sm.sum()
    assert np.allclose(post, sm, atol=1e-5) and abs(ll - np.log(np.exp(per).sum())) < 1e-4
    for i in range(10):
        assert abs(g.component_log_likelihood(x, i) - per[i]) < 1e-4
    g2 = khg.DiagGmm(nmix=2, dim=8)
    with pytest.raises(khg.KhgError):        # "Must call ComputeGconsts() before computing likelihood"
        g2.log_likelihood(x)
    with pytest.raises(khg.KhgError):        # dimension mismatch
        g.log_likelihood(x[:5])


def test_accumulate_from_diag_like_reference_test(khg):
    # python/tests/test_mle_diag_gmm.py:200-252
    rng = np.random.default_rng(7)
    g, w, mean, var = _gmm(khg, rng, 10, 8)
    x = rng.random(8).astype(np.float32)
    ll0, post = g.component_posteriors(x)
    for weight in (1.0, 0.25):
        acc = khg.AccumDiagGmm(g, khg.GmmUpdateFlags.kGmmAll)
        ll = acc.accumulate_from_diag(gmm=g, data=x, weight=weight)
        assert abs(ll - ll0) < 1e-5
        assert np.allclose(acc.occupancy, post * weight, rtol=1e-5)
        assert np.allclose(acc.mean_accumulator, np.outer(post * weight, x), rtol=1e-5, atol=1e-7)
        assert np.allclose(acc.variance_accumulator, np.outer(post * weight, x * x), rtol=1e-5, atol=1e-7)


def test_accumulate_for_gmm_two_feats(khg):
    # csrc/mle-am-diag-gmm.cc:54-76: posteriors of data1, statistics of data2
    rng = np.random.default_rng(8)
    g, w, mean, var = _gmm(khg, rng, 6, 8)
    am = khg.AmDiagGmm()
    am.add_pdf(g)
    x1, x2 = rng.random(8).astype(np.float32), rng.random(8).astype(np.float32)
    ll0, post = g.component_posteriors(x1)
    acc = khg.AccumAmDiagGmm()
    acc.init(am, khg.GmmUpdateFlags.kGmmAll)
    ll = acc.accumulate_for_gmm_two_feats(model=am, data1=x1, data2=x2, gmm_index=0, weight=0.5)
    assert abs(ll - ll0) < 1e-5
    a = acc.get_acc(0)
    assert np.allclose(a.occupancy, post * 0.5, rtol=1e-6)
    assert np.allclose(a.mean_accumulator, np.outer(post * 0.5, x2), rtol=1e-5, atol=1e-7)
    assert np.allclose(a.variance_accumulator, np.outer(post * 0.5, x2 * x2), rtol=1e-5, atol=1e-7)
    assert abs(acc.tot_count - 0.5) < 1e-7 and abs(acc.tot_log_like - 0.5 * ll0) < 1e-5


def test_invalid_model_raises_like_reference(khg):
    # decodable-am-diag-gmm.cc:63-65 / diag-gmm.cc:160-162: NaN/Inf log-likelihood -> RuntimeError
    g = khg.DiagGmm(nmix=2, dim=3)
    g.set_weights(np.array([0.0, 0.0], np.float32))     # gconsts = -inf for both: log-sum-exp = -inf
    g.set_means(np.zeros((2, 3), np.float32))
    assert g.compute_gconsts() == 2
    with pytest.raises(khg.KhgError):
        g.log_likelihood(np.zeros(3, np.float32))


def _mini_problem(khg, seed=3, n_utt=8):
    """3-state monophone model over 4 phones, linear transcripts, features drawn from the model."""
    rng = np.random.default_rng(seed)
    topo = khg.HmmTopology()
    s = "<Topology> <TopologyEntry> <ForPhones> 1 2 3 4 </ForPhones> "
    for i in range(3):
        s += f"<State> {i} <PdfClass> {i} <Transition> {i} 0.75 <Transition> {i + 1} 0.25 </State> "
    s += "<State> 3 </State> </TopologyEntry> </Topology>"
    topo.read(s)
    D = 6
    allx = (rng.standard_normal((500, D)) * 3).astype(np.float32)
    tm, tree, am = khg.gmm_init_mono(topo, allx)
    # spread the 12 pdfs apart so alignment is well defined
    true_means = (rng.standard_normal((tm.num_pdfs, D)) * 4).astype(np.float32)
    for p in range(tm.num_pdfs):
        g = am.get_pdf(p); g.set_means(true_means[p][None, :]); g.compute_gconsts()
    utts = []
    for u in range(n_utt):
        phones = rng.integers(1, 5, size=int(rng.integers(2, 5)))
        fst = khg.StdVectorFst()
        st = fst.add_state(); fst.start = st
        feats, ali_ref = [], []
        prev_loop = None
        for ph in phones:
            for hs in range(3):
                pdf = 3 * (ph - 1) + hs
                loop_tid, fwd_tid = 2 * pdf + 1, 2 * pdf + 2
                nxt = fst.add_state()
                fst.add_arc(st, khg.StdArc(fwd_tid, int(ph) if hs == 0 else 0, 0.0, nxt))
                fst.add_arc(nxt, khg.StdArc(loop_tid, 0, 0.0, nxt))
                d = int(rng.integers(1, 5))
                feats.append(true_means[pdf] + rng.standard_normal((d, D)).astype(np.float32))
                ali_ref += [fwd_tid] + [loop_tid] * (d - 1)
                st = nxt
        fst.set_final(st, 0.0)
        utts.append((f"utt{u}", fst, np.concatenate(feats).astype(np.float32), ali_ref, [int(p) for p in phones]))
    return topo, tm, tree, am, utts


def test_gmm_align_compiled_and_acc_stats_and_est_vs_oracle(khg):
    """One EM iteration through the reference's four entry points, single-utterance and batched,
    against the oracle pipeline (AddTransitionProbs -> AlignUtteranceWrapper -> acc-stats -> M-step)."""
    topo, tm, tree, am, utts = _mini_problem(khg)
    go, gc, w, miv, iv = am.flat()
    om = orc.OModel(go, gc, miv, iv)
    id2pdf = np.asarray(tm.transition_id_to_pdf_array(), np.int32)
    cfg = khg.AlignConfig(beam=200.0, retry_beam=0.0)
    num_done = num_err = num_retried = frame_count = 0
    tot_like = 0.0
    accs = khg.AccumAmDiagGmm(); accs.init(am, khg.GmmUpdateFlags.kGmmAll)
    tacc = None
    oacc = orc.OAccs(int(go[-1]), am.dim, tm.num_transition_ids)
    alis = []
    for name, fst, feats, ali_ref, words in utts:
        f2 = fst.copy()
        r = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt=name, fst=f2, feats=feats, align_config=cfg,
                                   acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1, num_done=num_done,
                                   num_error=num_err, num_retried=num_retried, tot_like=tot_like, frame_count=frame_count)
        num_done, num_err, num_retried, tot_like, frame_count = (r["num_done"], r["num_error"], r["num_retried"], r["tot_like"],
                                                                  r["frame_count"])
        c = f2.to_csr()    # f2 now carries the transition probs, like the reference's mutated copy
        og = orc.OGraph(c["start"], c["arc_off"], c["ilabel"], c["olabel"], c["weight"], c["nextstate"], c["final"])
        want = orc.align_utterance(og, om, id2pdf, feats, acoustic_scale=0.1)
        assert want["status"] == 0 and r["alignment"] == want["ali"].tolist() == ali_ref
        assert r["words"] == want["words"].tolist() == words
        ll, tacc = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=feats, ali=r["alignment"],
                                         transition_accs=tacc)
        oll = orc.acc_stats_ali(om, id2pdf, feats, want["ali"], oacc)
        assert ll == pytest.approx(oll, rel=1e-5)
        alis.append(r["alignment"])
    assert num_done == len(utts) and num_err == 0 and frame_count == sum(len(u[3]) for u in utts)
    assert tacc.sum() == frame_count                       # scripts/test_gmm_acc_stats_ali.py:106
    assert np.array_equal(tacc, oacc.trans_acc)
    assert accs.tot_count == frame_count and accs.tot_log_like == pytest.approx(oacc.total_log_like, rel=1e-5)
    # batched variant == loop of single calls
    rb = khg.gmm_align_compiled_batch(am, tm, [u[0] for u in utts], [u[1] for u in utts], [u[2] for u in utts], cfg,
                                      acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
    assert rb["alignment"] == alis and rb["num_done"] == num_done and rb["frame_count"] == frame_count
    assert rb["tot_like"] == pytest.approx(tot_like, rel=1e-6)
    accs_b = khg.AccumAmDiagGmm(); accs_b.init(am, khg.GmmUpdateFlags.kGmmAll)
    llb, tacc_b = khg.gmm_acc_stats_ali_batch(am, accs_b, tm, [u[2] for u in utts], alis)
    assert np.array_equal(tacc_b, tacc)
    for p in range(am.num_pdfs):
        np.testing.assert_allclose(accs_b.get_acc(p).occupancy, accs.get_acc(p).occupancy, rtol=1e-12)
        np.testing.assert_allclose(accs.get_acc(p).mean_accumulator, oacc.mean_acc[go[p]: go[p + 1]], rtol=2e-5, atol=1e-5)
    # M-step + transition update, then the model must still evaluate (gconsts valid) and have moved
    before = am.get_pdf(0).means.copy()
    info = khg.gmm_est(am, accs, tm, tacc, khg.MleTransitionUpdateConfig(), khg.MleDiagGmmOptions(min_gaussian_occupancy=3),
                       mixup=0, update_flags="mvwt", verbose=False)
    assert info["gmm_count"] == pytest.approx(frame_count, rel=1e-6) and np.isfinite(info["gmm_objf_impr"])
    assert not np.allclose(am.get_pdf(0).means, before) or accs.get_acc(0).occupancy.sum() <= 3
    assert np.isfinite(am.get_pdf(1).log_likelihood(utts[0][2][0]))


def _dev(ctx, m, gc, ut, cost, graphs=None):
    from kaldi_hmm_gmm_amd import DeviceModel, DeviceTransitions, UtteranceSet

    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=graphs if graphs is not None else ut.graphs)
    return dm, tm, us


def test_fallback_kernel_is_exercised_and_exact(ctx):
    """Long chains (> min_active live states) with a tiny beam defeat the beam certificate, so the
    order-faithful FasterDecoder kernel decides; it must equal the oracle token for token."""
    m, gc, om, ut, cost = build(90, 2, 10, n_utt=24, seed=21, min_phones=8, max_phones=20)
    # weak models (shared means) so that the best path wanders and the beam really prunes
    dm, tm, us = _dev(ctx, m, gc, ut, cost)
    poff, pdfs = us.pdf_lists()
    rng = np.random.default_rng(0)
    mats = [(-8 * rng.random((poff[u + 1] - poff[u], int(ut.frame_off[u + 1] - ut.frame_off[u])))).astype(np.float32)
            for u in range(us.n_utt)]
    us.upload_loglikes(mats)
    seen_fallback = 0
    for beam, retry in ((0.8, 4.0), (2.5, 0.0), (200.0, 0.0)):
        res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=1.0)
        for u in range(us.n_utt):
            T = int(ut.frame_off[u + 1] - ut.frame_off[u])
            want = orc.align_utterance_ll(oracle_graph(ut, u, cost), m.id2pdf, T, pdfs[poff[u]: poff[u + 1]], mats[u],
                                          acoustic_scale=1.0, beam=beam, retry_beam=retry)
            st = int(res["status"][u])
            seen_fallback += (st & 8) != 0
            assert (st & 3) == (want["status"] & 3), (beam, u, st, want["status"])
            a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
            if want["status"] & 1:
                assert (a == 0).all()
            else:
                assert (a == want["ali"]).all(), (beam, u)
                assert res["like"][u] == pytest.approx(want["like"], rel=1e-6, abs=1e-4)
    assert seen_fallback > 0


def test_epsilon_arcs_words_careful_and_errors(ctx):
    """Generic (non register-resident) DP path: epsilon-input arcs with word labels, branches,
    a graph whose final state is unreachable, an empty graph, a zero-frame utterance."""
    from kaldi_hmm_gmm_amd import synth

    rng = np.random.default_rng(5)
    m = synth.make_model(12, 3, 8, seed=5)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    graphs = [random_graph(rng, m.num_tids, n_main=int(rng.integers(3, 9)), p_eps=0.5) for _ in range(10)]
    graphs.append(random_graph(rng, m.num_tids, n_main=5, with_final=False))        # no final state at all
    graphs.append({"start": -1, "arc_off": np.zeros(1, np.int64), "ilabel": np.zeros(0, np.int32),
                   "olabel": np.zeros(0, np.int32), "weight": np.zeros(0, np.float32), "nextstate": np.zeros(0, np.int32),
                   "final": np.zeros(0, np.float32)})                                 # empty FST
    T = [int(rng.integers(len(g["final"]) + 1, 30)) for g in graphs]
    T[3] = 2                                                                          # too short to reach the final
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    feats = (rng.standard_normal((frame_off[-1], 8)) * 3).astype(np.float32)

    class UT:
        pass
    ut = UT(); ut.frame_off = frame_off; ut.feats = feats; ut.graphs = concat(graphs)
    cost = np.zeros(m.num_tids + 1, np.float32)
    dm, tm, us = _dev(ctx, m, gc, ut, cost)
    us.loglikes(dm)
    lls = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    for beam, retry in ((200.0, 0.0), (3.0, 9.0)):
        res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.3)
        for u, g in enumerate(graphs):
            og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
            want = orc.align_utterance_ll(og, m.id2pdf, T[u], pdfs[poff[u]: poff[u + 1]], lls[u] if lls[u].size else
                                          np.zeros((1, max(T[u], 1)), np.float32), acoustic_scale=0.3, beam=beam, retry_beam=retry)
            st = int(res["status"][u])
            assert (st & 1) == (want["status"] & 1), (u, st, want["status"])
            if g["start"] >= 0:
                assert (st & 2) == (want["status"] & 2), (u, st, want["status"])
            a = res["ali"][frame_off[u]: frame_off[u + 1]]
            if want["status"] & 1:
                assert (a == 0).all()
            else:
                assert (a == want["ali"]).all(), (beam, u)
                w = res["words"][res["words_off"][u]: res["words_off"][u + 1]]
                assert (w == want["words"]).all()
                assert res["like"][u] == pytest.approx(want["like"], rel=1e-5, abs=1e-4)
    assert int(res["status"][10]) & 1 and int(res["status"][11]) & 1


def test_exact_ties_follow_reference_token_order(ctx):
    """Two parallel branches with identical labels and weights tie exactly; the reference keeps the
    first-inserted token (strict '<', faster-decoder.cc:218-227).  Ties void the DP certificate and the
    faithful kernel must reproduce the oracle's choice."""
    from kaldi_hmm_gmm_amd import synth

    m = synth.make_model(6, 2, 5, seed=2)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    # state 0 -(tid 2)-> 1 and 0 -(tid 2)-> 2 (same label, same weight), both -> 3 with tid 4, loops everywhere
    arcs = [(0, 2, 11, 0.0, 1), (0, 2, 22, 0.0, 2), (1, 1, 0, 0.0, 1), (1, 4, 0, 0.0, 3), (2, 1, 0, 0.0, 2), (2, 4, 0, 0.0, 3),
            (3, 3, 0, 0.0, 3)]
    S = 4
    off = np.zeros(S + 1, np.int64)
    for a in arcs:
        off[a[0] + 1] += 1
    g = {"start": 0, "arc_off": np.cumsum(off), "ilabel": np.array([a[1] for a in arcs], np.int32),
         "olabel": np.array([a[2] for a in arcs], np.int32), "weight": np.array([a[3] for a in arcs], np.float32),
         "nextstate": np.array([a[4] for a in arcs], np.int32), "final": np.array([np.inf, np.inf, np.inf, 0.0], np.float32)}
    rng = np.random.default_rng(1)
    T = 9
    feats = (rng.standard_normal((T, 5)) * 2).astype(np.float32)

    class UT:
        pass
    ut = UT(); ut.frame_off = np.array([0, T], np.int64); ut.feats = feats; ut.graphs = concat([g])
    dm, tm, us = _dev(ctx, m, gc, ut, np.zeros(m.num_tids + 1, np.float32))
    us.loglikes(dm)
    res = us.align(tm, acoustic_scale=1.0)
    poff, pdfs = us.pdf_lists()
    og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
    want = orc.align_utterance_ll(og, m.id2pdf, T, pdfs, us.download_loglikes()[0], acoustic_scale=1.0)
    assert want["status"] == 0 and int(res["status"][0]) & 8, "a tie must route through the faithful kernel"
    assert (res["ali"] == want["ali"]).all() and (res["words"] == want["words"]).all()


def test_bad_configs_raise(ctx):
    m, gc, om, ut, cost = build(12, 2, 8, n_utt=2, seed=1)
    dm, tm, us = _dev(ctx, m, gc, ut, cost)
    import kaldi_hmm_gmm_amd as khg
    with pytest.raises(khg.KhgError):          # khg_align before khg_loglikes
        us.align(tm)
    us.loglikes(dm)
    with pytest.raises(khg.KhgError):          # decoder-wrappers.cc:29-33
        us.align(tm, beam=10.0, retry_beam=5.0)
    with pytest.raises(khg.KhgError):
        us.align(tm, beam=0.0)
    bad = dict(ut.graphs); bad["ilabel"] = bad["ilabel"].copy(); bad["ilabel"][0] = m.num_tids + 5
    with pytest.raises(khg.KhgError):          # hmm-utils.cc:484-488 invalid symbol on graph input side
        from kaldi_hmm_gmm_amd import UtteranceSet
        UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=bad)


def test_faster_decoder_class_like_reference_binding(khg):
    """python/csrc/faster-decoder.cc:33-53: decode / reached_final / get_best_path on the GPU path vs the
    oracle's FasterDecoder (best path labels and the LatticeWeight total of csrc/decoder-wrappers.cc:93-95)."""
    topo, tm, tree, am, utts = _mini_problem(khg)
    go, gc, w, miv, iv = am.flat()
    om = orc.OModel(go, gc, miv, iv)
    id2pdf = np.asarray(tm.transition_id_to_pdf_array(), np.int32)
    opts = khg.FasterDecoderOptions(beam=16.0)
    for name, fst, feats, ali_ref, words in utts[:3]:
        g = fst.copy()
        khg.add_transition_probs(tm, [], 1.0, 0.1, g)
        dec = khg.FasterDecoder(g, opts)
        assert dec.num_frames_decoded() == -1
        dec.decode(khg.DecodableAmDiagGmmScaled(am, tm, feats, 0.1))
        assert dec.reached_final() and dec.num_frames_decoded() == feats.shape[0]
        ok, lat = dec.get_best_path()
        assert ok and lat.num_states == feats.shape[0] + 1
        ok, il, ol, wt = lat.get_linear_symbol_sequence()
        c = g.to_csr()
        og = orc.OGraph(c["start"], c["arc_off"], c["ilabel"], c["olabel"], c["weight"], c["nextstate"], c["final"])
        want = orc.align_utterance(og, om, id2pdf, feats, acoustic_scale=0.1, beam=16.0)
        assert il == want["ali"].tolist() == ali_ref and ol == want["words"].tolist()
        assert -(wt.value1 + wt.value2) / 0.1 == pytest.approx(want["like"], rel=1e-5)
    # a graph whose final state cannot be reached in T frames: no best path on this path
    name, fst, feats, _, _ = utts[0]
    dec = khg.FasterDecoder(fst.copy(), opts)
    dec.decode(khg.DecodableAmDiagGmmScaled(am, tm, feats[:2], 0.1))
    assert not dec.reached_final() and dec.get_best_path()[0] is False
    with pytest.raises(khg.KhgError):
        khg.FasterDecoder(fst, khg.FasterDecoderOptions(max_active=1))


def test_wave_parallel_faithful_decoder_equals_serial_and_oracle(ctx, opt):
    """The wave-parallel order-faithful decoders -- the chain form (one lane per token, DPP prefix-min, parked slots) and the general
    wave form (prefix-min pruning, atomic per-state minima, first-insertion list order) -- against the one-lane emulation and the oracle's FasterDecoder, on scores that make the beam
    really prune, for the GetCutoff branches (default, max_active, min_active = 0)."""
    m, gc, om, ut, cost = build(120, 2, 10, n_utt=40, seed=33, min_phones=8, max_phones=30)
    dm, tm, us = _dev(ctx, m, gc, ut, cost)
    poff, pdfs = us.pdf_lists()
    rng = np.random.default_rng(5)
    mats = [(-6 * rng.random((poff[u + 1] - poff[u], int(ut.frame_off[u + 1] - ut.frame_off[u])))).astype(np.float32)
            for u in range(us.n_utt)]
    us.upload_loglikes(mats)
    seen = 0
    for kw in (dict(beam=1.5, retry_beam=6.0), dict(beam=3.0, retry_beam=0.0, max_active=12, min_active=3),
               dict(beam=2.0, retry_beam=8.0, min_active=0), dict(beam=4.0, retry_beam=0.0, max_active=40, min_active=20, beam_delta=0.25)):
        opt("k2_serial", 0)                   # the chain form (these graphs: no epsilon arcs, out-degree 2)
        rw = us.align(tm, acoustic_scale=1.0, **kw)
        opt("k2_serial", 3)                   # the general wave form
        rg = us.align(tm, acoustic_scale=1.0, **kw)
        opt("k2_serial", 1)                   # the one-lane emulation
        rs = us.align(tm, acoustic_scale=1.0, **kw)
        for r in (rw, rg):
            assert np.array_equal(r["status"], rs["status"]), kw
            assert np.array_equal(r["ali"], rs["ali"]), kw
            np.testing.assert_array_equal(r["like"], rs["like"])
        seen += int(((rw["status"] & 8) != 0).sum())
        for u in range(0, us.n_utt, 5):
            T = int(ut.frame_off[u + 1] - ut.frame_off[u])
            want = orc.align_utterance_ll(oracle_graph(ut, u, cost), m.id2pdf, T, pdfs[poff[u]: poff[u + 1]], mats[u],
                                          acoustic_scale=1.0, **kw)
            assert (int(rw["status"][u]) & 3) == (want["status"] & 3), (kw, u)
            a = rw["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
            if not (want["status"] & 1):
                assert (a == want["ali"]).all(), (kw, u)
    assert seen > 20, "the fallback decoder was hardly exercised"


def test_wave_faithful_decoder_epsilon_arcs_and_wide_fanout_equal_serial_and_oracle(ctx, opt):
    """Graphs with epsilon-input arcs (chains, equal-cost parallel epsilon paths, word labels) and a start state of
    out-degree 20: they have no beam certificate, so the order-faithful decoder IS their decoder.  The wave-parallel
    form (exact slot prefix sums for any out-degree; ProcessNonemitting as a lane-0 worklist over the epsilon-capable
    tokens, faster-decoder.cc:58-118) against the one-lane emulation and the oracle's FasterDecoder, token for token,
    with beams that really prune and all GetCutoff branches."""
    from kaldi_hmm_gmm_amd import synth

    rng = np.random.default_rng(77)
    m = synth.make_model(30, 2, 6, seed=9)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    graphs = [hub_graph(rng, m.num_tids, fan=int(rng.integers(9, 24)), tail=int(rng.integers(3, 9)), eps_ties=bool(i % 2)) for i in range(12)]
    graphs += [random_graph(rng, m.num_tids, n_main=int(rng.integers(5, 60)), p_eps=float(rng.choice([0.3, 0.8])), p_branch=0.5, p_long=0.5)
               for _ in range(28)]
    T = [int(rng.integers(8, 60)) for g in graphs]
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    feats = (rng.standard_normal((frame_off[-1], 6)) * 3).astype(np.float32)

    class UT:
        pass
    ut = UT(); ut.frame_off = frame_off; ut.feats = feats; ut.graphs = concat(graphs)
    cost = np.zeros(m.num_tids + 1, np.float32)
    dm, tm, us = _dev(ctx, m, gc, ut, cost)
    poff, pdfs = us.pdf_lists()
    # scores on a coarse grid: exact cost ties between different paths really occur
    mats = [(-0.25 * rng.integers(0, 24, size=(poff[u + 1] - poff[u], T[u]))).astype(np.float32) for u in range(us.n_utt)]
    us.upload_loglikes(mats)
    seen = 0
    for kw in (dict(beam=200.0, retry_beam=0.0), dict(beam=1.5, retry_beam=6.0), dict(beam=3.0, retry_beam=0.0, max_active=12, min_active=3),
               dict(beam=2.0, retry_beam=8.0, min_active=0), dict(beam=4.0, retry_beam=0.0, max_active=40, min_active=20, beam_delta=0.25)):
        opt("k2_serial", 0)
        rw = us.align(tm, acoustic_scale=1.0, **kw)
        opt("k2_serial", 1)
        rs = us.align(tm, acoustic_scale=1.0, **kw)
        assert np.array_equal(rw["status"], rs["status"]), kw
        assert np.array_equal(rw["ali"], rs["ali"]), kw
        assert np.array_equal(rw["words"], rs["words"]) and np.array_equal(rw["words_off"], rs["words_off"]), kw
        np.testing.assert_array_equal(rw["like"], rs["like"])
        seen += int(((rw["status"] & 8) != 0).sum())
        for u, g in enumerate(graphs):
            og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
            want = orc.align_utterance_ll(og, m.id2pdf, T[u], pdfs[poff[u]: poff[u + 1]], mats[u], acoustic_scale=1.0, **kw)
            assert (int(rw["status"][u]) & 3) == (want["status"] & 3), (kw, u)
            a = rw["ali"][frame_off[u]: frame_off[u + 1]]
            if not (want["status"] & 1):
                assert (a == want["ali"]).all(), (kw, u)
                w = rw["words"][rw["words_off"][u]: rw["words_off"][u + 1]]
                assert (w == want["words"]).all(), (kw, u)
    assert seen > 100, "the order-faithful decoder was hardly exercised"


@pytest.mark.parametrize("sizes", [((1100, 0.0), (1500, 0.0), (1300, 0.2), (40, 0.0)), ((2300, 0.0), (1900, 0.3), (1200, 0.0))])
def test_wave_faithful_decoder_above_1000_states_shared_hash_buckets(ctx, opt, sizes):
    """Graphs of more than 1000 states with state numbers in random order: states s and s + k hash_size share a bucket of the
    reference's HashList, so the token list is no longer in insertion order (hash-list-inl.h:129-174; the hash size follows
    faster-decoder.cc:337-344 and never shrinks).  The wave-parallel decoder (all tables in LDS; graph tables in HBM scratch --
    the second set only fits that way, the first is forced into it with k2_serial = 2) against the one-lane emulation and the
    oracle's FasterDecoder, with every state alive (max_active set: no beam certificate) and with beams that prune."""
    from kaldi_hmm_gmm_amd import synth

    rng = np.random.default_rng(1234 + len(sizes))
    m = synth.make_model(30, 2, 6, seed=9)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    graphs = [permute_states(random_graph(rng, m.num_tids, n_main=n, p_eps=pe, p_branch=0.4), rng) for n, pe in sizes]
    for g in graphs:          # weights and scores on one coarse grid: equal-cost tokens everywhere, so WHICH path wins depends on the list order
        g["weight"] = (np.round(g["weight"] * 4) / 4).astype(np.float32)
    T = [int(1.45 * n) + int(rng.integers(10, 120)) for n, _ in sizes]
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    feats = (rng.standard_normal((frame_off[-1], 6)) * 3).astype(np.float32)

    class UT:
        pass
    ut = UT(); ut.frame_off = frame_off; ut.feats = feats; ut.graphs = concat(graphs)
    cost = np.zeros(m.num_tids + 1, np.float32)
    dm, tm, us = _dev(ctx, m, gc, ut, cost)
    poff, pdfs = us.pdf_lists()
    mats = [(-0.25 * rng.integers(0, 24, size=(poff[u + 1] - poff[u], T[u]))).astype(np.float32) for u in range(us.n_utt)]
    us.upload_loglikes(mats)
    seen = 0
    ok = 0
    for ki, kw in enumerate((dict(beam=200.0, retry_beam=0.0, max_active=100000), dict(beam=40.0, retry_beam=0.0, max_active=700, min_active=3),
                             dict(beam=20.0, retry_beam=60.0, min_active=0), dict(beam=60.0, retry_beam=0.0, max_active=100000, hash_ratio=1.0))):
        res = {}
        modes = (0, 2, 1) if ki in (1, 2) else (0, 2)       # (the one-lane emulation -- seconds per call at these sizes -- on the two pruning configurations; the oracle checks all four)
        for mode in modes:
            opt("k2_serial", mode)
            res[mode] = us.align(tm, acoustic_scale=1.0, **kw)
        rw = res[0]
        for mode in modes[1:]:
            assert np.array_equal(rw["status"], res[mode]["status"]), (kw, mode)
            assert np.array_equal(rw["ali"], res[mode]["ali"]), (kw, mode)
            assert np.array_equal(rw["words"], res[mode]["words"]) and np.array_equal(rw["words_off"], res[mode]["words_off"]), (kw, mode)
            np.testing.assert_array_equal(rw["like"], res[mode]["like"])
        seen += int(((rw["status"] & 8) != 0).sum())
        for u, g in enumerate(graphs):
            og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
            want = orc.align_utterance_ll(og, m.id2pdf, T[u], pdfs[poff[u]: poff[u + 1]], mats[u], acoustic_scale=1.0, **kw)
            assert (int(rw["status"][u]) & 3) == (want["status"] & 3), (kw, u)
            a = rw["ali"][frame_off[u]: frame_off[u + 1]]
            if not (want["status"] & 1):
                ok += 1
                assert (a == want["ali"]).all(), (kw, u)
                w = rw["words"][rw["words_off"][u]: rw["words_off"][u + 1]]
                assert (w == want["words"]).all(), (kw, u)
    assert seen >= 3 * len(sizes) and ok >= 3 * len(sizes), "the order-faithful decoder was hardly exercised"
    for o in (us, tm, dm):
        o.close()


def test_accs_as_torch_aliases_the_device_block_and_all_reduces(ctx):
    """DeviceAccs.as_torch(): the zero-copy torch view bench.py / ResidentEm hand to torch.distributed.all_reduce
    (RCCL).  One-rank process group: the collective runs on the aliased memory and leaves the sums in place."""
    import os
    import torch
    import torch.distributed as dist
    from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet

    m, gc, om, ut, cost = build(12, 8, 16, n_utt=6, seed=5, max_phones=4)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats)
    us.upload_ali(ut.ref_ali)
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs)
    before = accs.download()
    ctx.sync()
    t = accs.as_torch()
    assert t.dtype == torch.float64 and t.numel() == accs.size and t.is_cuda and t.data_ptr() == accs.device_ptr()
    assert float(t[: accs.sumG].sum()) == pytest.approx(before["occ"].sum(), rel=1e-12)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.mul_(2.0)                                     # visible through the C-ABI download: same memory
        torch.cuda.synchronize()
        after = accs.download()
        np.testing.assert_array_equal(after["occ"], 2.0 * before["occ"])
        np.testing.assert_array_equal(after["mean_acc"], 2.0 * before["mean_acc"])
        assert after["total_frames"] == 2.0 * before["total_frames"]
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("beam,retry", [(200.0, 0.0), (6.0, 30.0)])
def test_graphs_too_large_for_lds_decode_from_hbm_scratch(ctx, beam, retry):
    """A decoding graph of thousands of states (tables beyond the 160 KB of LDS: khg_align used to refuse it) runs the
    generic DP and the order-faithful decoder out of an HBM scratch slice; status / alignment / words equal the oracle's
    FasterDecoder on the same scores.  One graph has epsilon-input arcs with word labels (no beam certificate path)."""
    from kaldi_hmm_gmm_amd import DeviceModel, DeviceTransitions, UtteranceSet, synth

    rng = np.random.default_rng(77)
    m = synth.make_model(30, 3, 8, seed=5)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    graphs = [random_graph(rng, m.num_tids, n_main=3000, p_branch=0.3, p_eps=0.0),
              random_graph(rng, m.num_tids, n_main=2600, p_branch=0.5, p_eps=0.15, p_long=0.3),
              random_graph(rng, m.num_tids, n_main=40, p_branch=0.3, p_eps=0.0)]
    T = [3100, 2700, 90]
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    feats = (rng.standard_normal((int(frame_off[-1]), 8)) * 2.0).astype(np.float32)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(np.zeros(m.num_tids + 1, np.float32))
    us = UtteranceSet(ctx, tm, frame_off, feats, graphs=concat(graphs))
    us.loglikes(dm)
    lls = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.3)
    for u, g in enumerate(graphs):
        og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
        want = orc.align_utterance_ll(og, m.id2pdf, T[u], pdfs[poff[u]: poff[u + 1]], lls[u], acoustic_scale=0.3, beam=beam, retry_beam=retry)
        st = int(res["status"][u])
        assert (st & 3) == (want["status"] & 3), (u, st, want["status"])
        a = res["ali"][frame_off[u]: frame_off[u + 1]]
        if want["status"] & 1:
            assert (a == 0).all()
        else:
            assert np.array_equal(a, want["ali"]), u
            assert np.array_equal(res["words"][res["words_off"][u]: res["words_off"][u + 1]], want["words"]), u
            assert abs(res["like"][u] - want["like"]) <= 1e-5 * abs(want["like"]) + 1e-3
    for o in (us, tm, dm):
        o.close()


def test_options_are_validated_and_round_trip(ctx):
    """khg_ctx_set_option / khg_ctx_get_option: names and numbers, range checks (KHG_E_ARG -> KhgError), previous value returned."""
    import kaldi_hmm_gmm_amd as khg

    old = ctx.set_option("k3_ny", 3)
    assert ctx.get_option("k3_ny") == 3 and ctx.get_option(13) == 3
    assert ctx.set_option(13, old) == 3 and ctx.get_option("k3_ny") == old
    for name, bad in (("k3_form", 3), ("k1_form", 6), ("k1_order", -1), ("k3_phase_b", 3)):
        before = ctx.get_option(name)
        with pytest.raises(khg.KhgError):
            ctx.set_option(name, bad)
        assert ctx.get_option(name) == before
    with pytest.raises(KeyError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(khg.KhgError):
        ctx.set_option(99, 1)
    ctx.set_k1_form("f16x2")
    assert ctx.get_option("k1_form") == 4
    ctx.set_k1_form("auto")
    assert ctx.get_option("k1_form") == 0
