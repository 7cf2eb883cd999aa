"""The reference EM driver's call sequence, with the reference's exact keyword arguments, on this package.

Restated (not copied) from egs/yesno/train.py:38-222: get the topology, gmm_init_mono, build L.fst, TrainingGraphCompiler(
trans_model=, ctx_dep=, lex_fst=, disambig_syms=, opts=), compile_graph_from_text, equal_align(ifst=, length=, rand_seed=,
num_retries=), AccumAmDiagGmm().init(model=, flags=), gmm_acc_stats_ali(am_gmm=, ...), gmm_est(...), then the realign loop with
gmm_boost_silence / gmm_align_compiled.  lhotse cuts and the yesno audio are replaced by synthetic YES/NO features.

Also here: every keyword the in-scope reference bindings name (python/csrc/*.cc) is CALLED once with that keyword -- a signature
that parses but does not bind would pass tests/test_pybind_signatures.py and fail here -- and align_utterance_wrapper /
FasterDecoder accept any DecodableInterface, including one written in Python (python/csrc/decodable-itf.cc:16-53)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIL, Y, N = 1, 2, 3            # phones
YES, NO = 2, 1                 # words, sorted like Lexiconp.word2id: <eps> NO YES


@pytest.fixture()
def khg(ctx):
    import kaldi_hmm_gmm_amd as k
    from kaldi_hmm_gmm_amd import _gpu
    _gpu.set_default_context(ctx)
    return k


def _cuts(n):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_mono_synthetic as ex
    rng = np.random.default_rng(11)
    utts = ex.make_data(n, 23, rng)
    # ex.YES = 1 / ex.NO = 2 there; this file numbers words like the reference's sorted word table
    return [(name, [YES if w == ex.YES else NO for w in words], x) for name, words, x in utts]


def test_reference_driver_call_sequence(khg):
    cuts = _cuts(12)
    topo = khg.generate_hmm_topo(non_sil_phones=[Y, N], sil_phone=SIL)
    transition_model, tree, am = khg.gmm_init_mono(topo=topo, cuts=np.concatenate([c[2] for c in cuts[:10]]))
    info = khg.gmm_info(am, transition_model)
    num_gauss = info["number_of_gaussians"]
    assert info == {"number_of_phones": 3, "number_of_pdfs": 11, "number_of_transition_ids": transition_model.num_transition_ids,
                    "number_of_transition_states": 11, "feature_dimensition": 23, "number_of_gaussians": 11}
    total_gauss, max_iter_inc = 40, 4
    inc_gauss = (total_gauss - num_gauss) // max_iter_inc

    lex_fst = khg.make_lexicon_fst_with_silence({NO: [(1.0, [N])], YES: [(1.0, [Y])]}, sil_prob=0.5, sil_phone=SIL)
    training_graph_compiler_opts = khg.TrainingGraphCompilerOptions()
    gc = khg.TrainingGraphCompiler(trans_model=transition_model, ctx_dep=tree, lex_fst=lex_fst, disambig_syms=[4, 5],
                                   opts=training_graph_compiler_opts)
    train_graphs = {cid: gc.compile_graph_from_text(words) for cid, words, _ in cuts}

    ali = {}
    for cid, _, x in cuts:
        succeeded, aligned_seq = khg.equal_align(ifst=train_graphs[cid], length=x.shape[0], rand_seed=3, num_retries=10)
        assert succeeded
        ali[cid] = aligned_seq

    def accumulate():
        gmm_accs = khg.AccumAmDiagGmm()
        gmm_accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)           # egs/yesno/train.py:110-111
        transition_accs = None
        tot_log_like = 0.0
        for cid, _, x in cuts:
            log_like, transition_accs = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=gmm_accs, transition_model=transition_model, feats=x,
                                                              ali=ali[cid], transition_accs=transition_accs)
            tot_log_like += log_like
        assert transition_accs.sum() == sum(c[2].shape[0] for c in cuts)   # scripts/test_gmm_acc_stats_ali.py:106
        return gmm_accs, transition_accs, tot_log_like / len(cuts)

    gmm_accs, transition_accs, avg0 = accumulate()
    tcfg = khg.MleTransitionUpdateConfig()
    gmm_opts = khg.MleDiagGmmOptions()
    gmm_opts.min_gaussian_occupancy = 3
    khg.gmm_est(am_gmm=am, gmm_accs=gmm_accs, transition_model=transition_model, transition_accs=transition_accs, tcfg=tcfg,
                gmm_opts=gmm_opts, mixup=num_gauss, mixdown=0, perturb_factor=0.01, power=0.2, min_count=20.0, update_flags="mvwt")
    avgs = []
    for i in range(6):
        if i in (1, 2, 3, 4, 5):
            am_b = khg.gmm_boost_silence(am_gmm=am, transition_model=transition_model, silence_phones=[SIL], boost=1.0)
            align_config = khg.AlignConfig()
            align_config.beam = 6.0
            align_config.retry_beam = 40.0
            align_config.careful = False
            for cid, words, x in cuts:
                ans = khg.gmm_align_compiled(am_gmm=am_b, transition_model=transition_model, utt=cid, fst=train_graphs[cid].copy(),
                                             feats=x, align_config=align_config, acoustic_scale=0.1, transition_scale=1.0,
                                             self_loop_scale=0.1)
                assert set(ans) >= {"alignment", "words", "num_done", "num_error", "num_retried", "tot_like", "frame_count"}
                if ans["alignment"]:
                    ali[cid] = ans["alignment"]
                    assert ans["words"] == words and len(ans["alignment"]) == x.shape[0]
        gmm_accs, transition_accs, avg = accumulate()
        avgs.append(avg)
        khg.gmm_est(am_gmm=am, gmm_accs=gmm_accs, transition_model=transition_model, transition_accs=transition_accs, tcfg=tcfg,
                    gmm_opts=khg.MleDiagGmmOptions(), mixup=num_gauss, mixdown=0, perturb_factor=0.01, power=0.2, min_count=20.0,
                    update_flags="mvwt")
        if i < max_iter_inc:
            num_gauss += inc_gauss
    assert avgs[-1] > avg0 + 5.0 * np.mean([c[2].shape[0] for c in cuts]) * 0.2, (avg0, avgs)       # EM raised the likelihood
    assert am.num_gauss > 11
    info = khg.gmm_info(am_gmm=am, transition_model=transition_model)
    assert info["number_of_gaussians"] == am.num_gauss


def test_every_reference_keyword_binds(khg):
    """python/csrc/diag-gmm.cc:19-150, mle-diag-gmm.cc:60-157, mle-am-diag-gmm.cc:14-59, am-diag-gmm.cc:13-45: one call per keyword form."""
    rng = np.random.default_rng(5)
    nmix, dim = 4, 6
    w = rng.random(nmix).astype(np.float32); w /= w.sum()
    mean = rng.standard_normal((nmix, dim)).astype(np.float32)
    var = (rng.random((nmix, dim)) + 0.5).astype(np.float32)
    g = khg.DiagGmm(nmix=nmix, dim=dim)
    g.set_weights(w=w); g.set_means(m=mean); g.set_invvars(inv_vars=1 / var)
    g.set_invvars_and_means(inv_vars=1 / var, means=mean)
    g.set_component_weight(gauss=0, weight=float(w[0]))
    g.set_component_mean(gauss=1, mean=mean[1]); g.set_component_inv_var(gauss=1, inv_var=1 / var[1])
    assert np.allclose(g.get_component_mean(gauss=1), mean[1], atol=1e-6) and np.allclose(g.get_component_variance(gauss=1), var[1], rtol=1e-6)
    g.compute_gconsts()
    x = rng.standard_normal(dim).astype(np.float32)
    ll = g.log_likelihoods(data=x)
    assert abs(g.log_likelihood(data=x) - float(np.logaddexp.reduce(ll.astype(np.float64)))) < 1e-4
    assert np.allclose(g.log_likelihoods_matrix(data=x[None])[0], ll, atol=1e-5)
    assert np.allclose(g.log_likelihoods_preselect(data=x, indices=[2, 0]), ll[[2, 0]], atol=1e-5)
    loglike, post = g.component_posteriors(data=x)
    assert abs(post.sum() - 1) < 1e-5 and abs(loglike - g.log_likelihood(data=x)) < 1e-4
    assert abs(g.component_log_likelihood(data=x, comp_id=3) - ll[3]) < 1e-5
    g2 = khg.DiagGmm(gmm=g)
    g3 = khg.DiagGmm(); g3.copy_from_diag_gmm(diaggmm=g)
    assert np.array_equal(g2.means_invvars, g.means_invvars) and np.array_equal(g3.gconsts, g.gconsts)
    # DiagGmm(gmms=[(weight, gmm) ...]): the weighted concatenation, gconsts computed (csrc/diag-gmm.cc:68-101)
    cat = khg.DiagGmm(gmms=[(0.25, g), (0.75, g2)])
    assert cat.num_gauss == 2 * nmix and cat.valid_gconsts
    assert np.allclose(cat.weights, np.concatenate([0.25 * w, 0.75 * w]), rtol=1e-6) and np.array_equal(cat.inv_vars[nmix:], g.inv_vars)
    assert abs(cat.log_likelihood(data=x) - g.log_likelihood(data=x)) < 1e-4           # a mixture of two copies of one density
    with pytest.raises(Exception):
        khg.DiagGmm(gmms=[(0.0, g)])
    g.interpolate(rho=0.5, source=g2)                                                    # flags defaults to kGmmAll
    g.remove_component(gauss=3, renorm_weights=True)
    assert g.split(target_components=4, perturb_factor=0.01) == [int(np.argmax(g2.weights[:3]))]
    g.merge(target_components=3)
    assert g.num_gauss == 3

    acc = khg.AccumDiagGmm(gmm=g2, flags=khg.GmmUpdateFlags.kGmmAll)
    acc2 = khg.AccumDiagGmm(); acc2.resize(num_gauss=nmix, dim=dim, flags=khg.kGmmAll)
    acc.accumulate_from_diag(gmm=g2, data=x, weight=1.0)
    acc.accumulate_for_component(data=x, comp_index=0, weight=0.5)
    acc.accumulate_from_posteriors(data=x, gauss_posteriors=post)
    acc.add_stats_for_component(g=1, occ=1.0, x_stats=x.astype(np.float64), x2_stats=(x * x).astype(np.float64))
    acc2.add(scale=1.0, acc=acc); acc2.scale(f=0.5, flags=khg.kGmmAll)
    assert np.allclose(acc2.occupancy, 0.5 * acc.occupancy)
    acc2.smooth_stats(tau=1.0); acc2.smooth_with_accum(tau=1.0, src_acc=acc); acc2.smooth_with_model(tau=1.0, src_gmm=g2)
    acc2.set_zero(flags=khg.kGmmAll)
    assert khg.ml_objective(gmm=g2, diaggmm_acc=acc) == khg.ml_objective(g2, acc)
    objf, count, fl_e, fl_g, removed = khg.mle_diag_gmm_update(config=khg.MleDiagGmmOptions(min_gaussian_occupancy=0.1), diag_gmm_acc=acc,
                                                               flags=khg.kGmmWeights, gmm=khg.DiagGmm(gmm=g2))
    assert count > 0
    khg.map_diag_gmm_update(config=khg.MapDiagGmmOptions(mean_tau=10.0, variance_tau=50.0, weight_tau=10.0), diag_gmm_acc=acc,
                            flags=khg.kGmmMeans, gmm=khg.DiagGmm(gmm=g2))

    am = khg.AmDiagGmm()
    am.init(proto=g2, num_pdfs=2); am.add_pdf(gmm=g2)
    assert am.num_gauss_in_pdf(pdf_index=2) == nmix
    am.set_gaussian_mean(pdf_index=1, gauss_index=0, **{"in": mean[3]})
    assert np.allclose(am.get_gaussian_mean(pdf_index=1, gauss=0), mean[3], atol=1e-6)
    assert np.allclose(am.get_gaussian_variance(pdf_index=1, gauss=0), var[0], rtol=1e-6)
    am.compute_gconsts()
    assert abs(am.log_likelihood(pdf_index=0, data=x) - g2.log_likelihood(data=x)) < 1e-5
    other = khg.AmDiagGmm(); other.copy_from_am_diag_gmm(other=am)
    other.split_pdf(pdf_idx=0, target_components=nmix + 1, perturb_factor=0.01)
    other.split_by_count(state_occs=np.array([10, 20, 30], np.float32), target_components=3 * nmix + 3, perturb_factor=0.01, power=0.2, min_count=1.0)
    other.merge_by_count(state_occs=np.array([10, 20, 30], np.float32), target_components=3 * nmix, power=0.2, min_count=1.0)
    assert other.num_gauss == 3 * nmix

    accs = khg.AccumAmDiagGmm()
    accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    assert accs.num_accs == 3 and accs.dim == dim
    accs.init(model=am, dim=dim, flags=khg.GmmUpdateFlags.kGmmAll)
    assert accs.accumulate_for_gmm(model=am, data=x, gmm_index=0, weight=1.0) == pytest.approx(g2.log_likelihood(data=x), abs=1e-4)
    accs.accumulate_for_gmm_two_feats(model=am, data1=x, data2=x, gmm_index=1, weight=1.0)
    accs.accumulate_from_posteriors(model=am, data=x, gmm_index=2, weight=post)         # the reference names the posteriors `weight`
    accs.accumulate_for_gaussian(am=am, data=x, gmm_index=2, gauss_index=1, weight=2.0)
    assert accs.tot_count == pytest.approx(3.0, abs=1e-5) and accs.tot_stats_count == pytest.approx(5.0, abs=1e-5)
    a2 = khg.AccumAmDiagGmm(); a2.init(am, khg.kGmmAll); a2.add(scale=2.0, other=accs); a2.scale(scale=0.5)
    assert np.allclose(a2.get_acc(2).occupancy, accs.get_acc(2).occupancy)
    accs.set_zero(flags=khg.kGmmAll)
    assert accs.tot_stats_count == 0
    khg.map_am_diag_gmm_update(config=khg.MapDiagGmmOptions(), amdiag_gmm_acc=a2, flags=khg.kGmmMeans, am_gmm=am)
    # (last: the ML update may remove Gaussians that saw no data, after which the accumulators no longer fit the model)
    khg.mle_am_diag_gmm_update(config=khg.MleDiagGmmOptions(min_gaussian_occupancy=0.01), amdiag_gmm_acc=a2, flags=khg.kGmmWeights, am_gmm=am)
    t = khg.TransitionModelTuple(phone=1, hmm_state=0, forward_pdf=2, self_loop_pdf=2)
    assert t == khg.TransitionModelTuple(1, 0, 2, 2) and khg.TransitionModelTuple().phone == 0


def _task(khg, seed=3):
    cuts = _cuts(4)
    np.random.seed(seed)             # gmm_init_mono's perturbation draws from numpy's global generator
    topo = khg.generate_hmm_topo(non_sil_phones=[Y, N], sil_phone=SIL)
    tm, tree, am = khg.gmm_init_mono(topo=topo, cuts=np.concatenate([c[2] for c in cuts]), perturb_factor=0.5)
    gc = khg.TrainingGraphCompiler(tm, tree, {NO: [(1.0, [N])], YES: [(1.0, [Y])]}, sil_phone=SIL)
    return cuts, tm, am, gc


class _MatrixDecodable:
    """built lazily so that the base class comes from the package under test"""

    @staticmethod
    def make(khg, scores, calls):
        class PyDec(khg.DecodableInterface):            # python/csrc/decodable-itf.cc:16-53: four overridable methods
            def __init__(self):
                super().__init__()

            def log_likelihood(self, frame, index):
                calls[0] += 1
                return float(scores[frame, index])

            def is_last_frame(self, frame):
                return frame == scores.shape[0] - 1

            def num_frames_ready(self):
                return scores.shape[0]

            def num_indices(self):
                return scores.shape[1] - 1
        return PyDec()


def test_align_utterance_wrapper_takes_any_decodable(khg):
    """decoder-wrappers.cc:25-47 takes a DecodableInterface*: the GMM decodable with scale != acoustic_scale, the unmapped GMM
    decodable is refused only by the graph (its indices are pdfs), and a Python subclass scoring from a matrix must give the same
    alignment, words and like as the GMM decodable it copies."""
    cuts, tm, am, gc = _task(khg)
    cfg = khg.AlignConfig(beam=200.0, retry_beam=0.0, careful=False)
    for cid, words, x in cuts[:3]:
        fst = gc.compile_graph_from_text(words)
        khg.add_transition_probs(trans_model=tm, transition_scale=1.0, self_loop_scale=0.1, fst=fst)
        dec = khg.DecodableAmDiagGmmScaled(am=am, tm=tm, feats=x, scale=0.1)
        want = khg.align_utterance_wrapper(config=cfg, utt=cid, acoustic_scale=0.1, fst=fst.copy(), decodable=dec, num_done=0, num_error=0,
                                           num_retried=0, tot_like=0.0, frame_count=0)
        assert want[0] == 1 and want[1] == 0 and want[4] == x.shape[0] and want[6] == words
        # scale != acoustic_scale: same path (scores scaled by the decodable), like divided by acoustic_scale (decoder-wrappers.cc:95)
        got = khg.align_utterance_wrapper(cfg, cid, 0.2, fst.copy(), dec, 0, 0, 0, 0.0, 0)
        assert got[5] == want[5] and got[3] == pytest.approx(want[3] * 0.1 / 0.2, rel=1e-6)
        # a decodable written in Python over the same scores: log_likelihood(frame, tid) = scale * ll(frame, pdf(tid))
        T, ntid = x.shape[0], tm.num_transition_ids
        scores = np.zeros((T, ntid + 1), np.float32)
        for tid in range(1, ntid + 1):
            for t in range(T):
                scores[t, tid] = dec.log_likelihood(frame=t, index=tid)
        calls = [0]
        pydec = _MatrixDecodable.make(khg, scores, calls)
        got = khg.align_utterance_wrapper(config=cfg, utt=cid, acoustic_scale=0.1, fst=fst.copy(), decodable=pydec, num_done=5, num_error=1,
                                          num_retried=2, tot_like=1.5, frame_count=7)
        assert calls[0] > 0, "the Python decodable was never asked for a score"
        assert got[0] == 6 and got[1] == 1 and got[2] == 2 and got[4] == 7 + T
        assert got[5] == want[5] and got[6] == want[6]
        assert got[3] - 1.5 == pytest.approx(want[3], rel=1e-6)
        # FasterDecoder over the Python decodable == over the GMM decodable (faster-decoder.cc:38-51)
        opts = khg.FasterDecoderOptions(beam=200.0)
        d1, d2 = khg.FasterDecoder(fst=fst, config=opts), khg.FasterDecoder(fst=fst, config=opts)
        d1.decode(decodable=dec); d2.decode(decodable=pydec)
        assert d1.reached_final() and d2.reached_final() and d2.num_frames_decoded() == T
        ok1, lat1 = d1.get_best_path(use_final_probs=True)
        ok2, lat2 = d2.get_best_path()
        s1, s2 = lat1.get_linear_symbol_sequence(), lat2.get_linear_symbol_sequence()
        assert ok1 and ok2 and s1[1] == s2[1] == want[5] and s1[2] == s2[2]
        assert s1[3].value1 == pytest.approx(s2[3].value1, rel=1e-6) and s1[3].value2 == pytest.approx(s2[3].value2, rel=1e-6)
        assert -(s2[3].value1 + s2[3].value2) / 0.1 == pytest.approx(want[3], rel=1e-5)
    # an empty graph counts an error, whatever the decodable (decoder-wrappers.cc:35-41)
    r = khg.align_utterance_wrapper(cfg, "empty", 0.1, khg.StdVectorFst(), pydec, 0, 0, 0, 0.0, 0)
    assert r[:3] == (0, 1, 0) and r[5] == [] and r[6] == []
    # too few frames for the graph: no final state reached -> error, retried counted when retry_beam is set
    short = _MatrixDecodable.make(khg, scores[:2], [0])
    r = khg.align_utterance_wrapper(khg.AlignConfig(beam=6.0, retry_beam=40.0), "short", 0.1, fst.copy(), short, 0, 0, 0, 0.0, 0)
    assert r[0] == 0 and r[1] == 1 and r[2] == 1
