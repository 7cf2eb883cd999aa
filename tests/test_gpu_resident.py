"""ResidentEm (features / graphs / alignments / accumulators / model resident in HBM, K4 M-step) against the
per-call scripts (gmm_align_compiled_batch / gmm_acc_stats_ali_batch / gmm_est, host M-step): same alignments
every pass, same number of Gaussians, parameters equal to float rounding (K3's fp64 atomics are unordered and
gconsts go through logf; see tests/test_gpu_mstep.py)."""
import copy
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def _clone(am):
    """Exact copy, gconsts included (pickling / deepcopy recompute them on the host, like the reference's binding)."""
    import kaldi_hmm_gmm_amd as khg
    c = khg.AmDiagGmm()
    c.copy_from_am_diag_gmm(am)
    return c


def _randn(seed):
    rng = np.random.default_rng(seed)
    return lambda d: rng.standard_normal(d).astype(np.float32)


def test_resident_em_matches_per_call_scripts(ctx):
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd import _gpu
    from kaldi_hmm_gmm_amd.training_graph import TrainingGraphCompiler, equal_align, generate_hmm_topo
    import train_mono_synthetic as ex

    _gpu.set_default_context(ctx)
    rng = np.random.default_rng(11)
    utts = ex.make_data(20, 13, rng)
    names, feats = [u[0] for u in utts], [u[2] for u in utts]
    topo = generate_hmm_topo(non_sil_phones=[ex.Y, ex.N], sil_phone=ex.SIL)

    def fresh():
        tm, tree, am = khg.gmm_init_mono(topo, np.concatenate(feats[:10]))
        comp = TrainingGraphCompiler(tm, tree, {ex.YES: [(1.0, [ex.Y])], ex.NO: [(1.0, [ex.N])]}, sil_phone=ex.SIL, sil_prob=0.5)
        graphs = comp.compile_graphs_from_text([u[1] for u in utts])
        ali = []
        for g, x in zip(graphs, feats):
            ok, a = equal_align(g, x.shape[0], rand_seed=3, num_retries=10)
            assert ok
            ali.append(a)
        return tm, am, graphs, ali

    cfg = khg.AlignConfig(beam=6.0, retry_beam=40.0, careful=False)
    tcfg = khg.MleTransitionUpdateConfig()
    opts = khg.MleDiagGmmOptions(min_gaussian_occupancy=3)
    mix = [11, 16, 22, 22, 22]

    tm_b, am_b, graphs, ali = fresh()
    em = khg.ResidentEm(am_b, tm_b, graphs, feats, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1, ctx=ctx)
    em.set_alignments(ali)
    rn_b, rn_a = _randn(5), _randn(5)
    likes = []
    for it, target in enumerate(mix):
        if it > 0:
            # boost: device vs scripts/gmm_boost_silence.py on a host copy (gconsts through logf: <= 4 ulps)
            am_0 = _clone(em.sync_host())
            w_before = am_0.flat()[2].copy()
            am_h = khg.gmm_boost_silence(am_0, tm_b, [ex.SIL], boost=1.25)       # returns the boosted copy ...
            assert am_h is not am_0 and np.array_equal(am_0.flat()[2], w_before)   # ... and leaves its argument alone
            assert not np.array_equal(am_h.flat()[2], w_before)
            em.boost_silence([ex.SIL], boost=1.25)
            am_d = _clone(em.sync_host())
            _, gc_h, w_h, _, _ = am_h.flat()
            _, gc_d, w_d, _, _ = am_d.flat()
            assert np.array_equal(w_h, w_d) and (np.abs(gc_h - gc_d) <= 4 * np.spacing(np.abs(gc_h))).all()
            # align: the per-call script on the SAME parameters gives the same alignments, counters and likelihood
            r = em.align(cfg)
            ra = khg.gmm_align_compiled_batch(am_d, tm_b, names, graphs, feats, cfg, acoustic_scale=0.1, transition_scale=1.0,
                                              self_loop_scale=0.1)
            ali = em.alignments()
            assert ali == ra["alignment"] and r["num_error"] == ra["num_error"] == 0
            assert r["num_done"] == ra["num_done"] == len(utts) and r["frame_count"] == ra["frame_count"]
            assert r["num_retried"] == ra["num_retried"] and r["tot_like"] == pytest.approx(ra["tot_like"], rel=1e-6)
        # acc-stats + est: per-call scripts (host M-step) from the same model / alignments / random numbers
        am_a, tm_a = _clone(em.sync_host()), copy.deepcopy(tm_b)
        accs = khg.AccumAmDiagGmm(); accs.init(am_a, khg.GmmUpdateFlags.kGmmAll)
        ll, tacc = khg.gmm_acc_stats_ali_batch(am_a, accs, tm_a, feats, ali)
        info_a = khg.gmm_est(am_a, accs, tm_a, tacc, tcfg, opts, mixup=target, update_flags="mvwt", verbose=False, randn=rn_a)
        st = em.accumulate()
        assert st["total_frames"] == sum(f.shape[0] for f in feats) and st["total_log_like"] == pytest.approx(ll, rel=1e-6)
        info_b = em.update(tcfg, opts, mixup=target, update_flags="mvwt", randn=rn_b)
        am_b2 = em.sync_host()
        assert am_a.num_gauss == am_b2.num_gauss == em.num_gauss
        assert info_b["gmm_count"] == pytest.approx(info_a["gmm_count"], rel=1e-6)
        assert info_b["avg_like"] == pytest.approx(info_a["avg_like"], rel=1e-6)
        assert info_b["transition_objf_impr"] == pytest.approx(info_a["transition_objf_impr"], rel=1e-5, abs=1e-5)
        assert info_b["gmm_objf_impr"] == pytest.approx(info_a["gmm_objf_impr"], rel=1e-3, abs=1e-2)
        for p in range(am_a.num_pdfs):
            ga, gb = am_a.get_pdf(p), am_b2.get_pdf(p)
            assert ga.num_gauss == gb.num_gauss
            # K3's fp64 atomics are unordered: the two accumulator sets agree to ~1e-13, the parameters to float rounding
            np.testing.assert_allclose(gb.weights, ga.weights, rtol=1e-6)
            np.testing.assert_allclose(gb.inv_vars, ga.inv_vars, rtol=1e-5)
            np.testing.assert_allclose(gb.means_invvars, ga.means_invvars, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(gb.gconsts, ga.gconsts, rtol=1e-6, atol=1e-5)
        for t in range(1, tm_a.num_transition_ids + 1):
            assert tm_b.get_transition_log_prob(t) == tm_a.get_transition_log_prob(t)
        likes.append(info_b["avg_like"])
    assert em.num_gauss == 22 and likes[-1] > likes[0] + 1.0
    # mixing down: AmDiagGmm::MergeByCount on the device model (khg_model_merge) against gmm_est(mixdown=...) on the host
    am_a, tm_a = _clone(em.sync_host()), copy.deepcopy(tm_b)
    accs = khg.AccumAmDiagGmm(); accs.init(am_a, khg.GmmUpdateFlags.kGmmAll)
    ll, tacc = khg.gmm_acc_stats_ali_batch(am_a, accs, tm_a, feats, ali)
    khg.gmm_est(am_a, accs, tm_a, tacc, tcfg, opts, mixdown=14, update_flags="mvwt", verbose=False)
    em.accumulate()
    em.update(tcfg, opts, mixdown=14, update_flags="mvwt")
    am_b2 = em.sync_host()
    assert am_a.num_gauss == am_b2.num_gauss == em.num_gauss and em.num_gauss < 22
    for p in range(am_a.num_pdfs):
        ga, gb = am_a.get_pdf(p), am_b2.get_pdf(p)
        assert ga.num_gauss == gb.num_gauss
        np.testing.assert_allclose(gb.weights, ga.weights, rtol=1e-5)
        np.testing.assert_allclose(gb.inv_vars, ga.inv_vars, rtol=1e-4)
        np.testing.assert_allclose(gb.means_invvars, ga.means_invvars, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gb.gconsts, ga.gconsts, rtol=1e-5, atol=1e-4)
    r = em.align(cfg)                        # and the merged model aligns
    assert r["num_error"] == 0
    em.close()
