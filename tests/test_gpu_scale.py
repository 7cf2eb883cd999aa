"""One EM pass at the benchmark's model shape (5000 pdfs x 64 Gaussians x 40 dims, bench-like utterances of 10..40 phones).
The oracle replays the first 240 utterances (~72 k frames: a few seconds on the host's cores, orc_em_pass_mt_keep) and K2 / K3 must
give its alignments / statistics on them; the whole set of 1500 goes through size-independent properties:

  K1  the pdf-major and the utterance-major kernels (two independent tilings) agree within 2 float ulps on every cell
  K2  every alignment is an accepting path of its graph; the returned likelihood is that path's cost replayed on
      the host in the reference's token arithmetic; the path is never worse than the generating path (no pruning)
  K3  sum(occ) = frames, transition counts = histogram of the alignment, sum_g mean_acc = sum_t x_t and
      sum_g var_acc = sum_t x_t^2 (posteriors of a frame sum to one)
"""
import numpy as np
import pytest

from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_em_pass_properties_at_bench_shape(ctx, opt):
    P, G, D, U = 5000, 64, 40, 1500
    m = synth.make_model(P, G, D, seed=20230417)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    ut = synth.make_utts(m, U, seed=5)
    N = int(ut.frame_off[-1])
    il = np.arange(m.num_tids + 1, dtype=np.int32)
    cost = orc.add_transition_probs(il, np.zeros(m.num_tids + 1, np.float32), m.log_probs, m.non_self_loop_log_probs,
                                    m.id2state, m.is_self_loop, 1.0, 0.1)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)

    # ---- K1: two fp32-MFMA tilings, one answer; the bf16x3 form (the default) within the fp32 bound of them ----
    opt.k1("pdf")
    us.loglikes(dm)
    ll = us.download_loglikes()
    opt.k1("utt")
    us.loglikes(dm)
    ll_utt = us.download_loglikes()
    opt.k1("auto")
    us.loglikes(dm)
    ll_b = us.download_loglikes()
    from helpers import exact_loglikes
    poff0, pdfs0 = us.pdf_lists()
    # the DEFAULT K1 (f16x2s) and the fp32-MFMA form against an fp64 evaluation: EVERY cell (all listed pdfs x all frames) of 60
    # utterances spread over the set, at the tolerance of tests/test_gpu_parity.py
    worst_b = worst_f = 0.0
    for u in np.linspace(0, U - 1, 60).astype(int):
        pl = pdfs0[poff0[u]: poff0[u + 1]]
        exact, bound = exact_loglikes(m, gc, ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]], pl)
        tol = 1e-5 + 1e-6 * bound
        eb, ef = np.abs(ll_b[u] - exact), np.abs(ll[u] - exact)
        assert (eb <= tol).all() and (ef <= tol).all(), (u, float((eb / tol).max()), float((ef / tol).max()))
        worst_b, worst_f = max(worst_b, float((eb / bound).max())), max(worst_f, float((ef / bound).max()))
    print("max |err| / B vs fp64: default K1 %.2e, fp32-MFMA K1 %.2e" % (worst_b, worst_f))
    # everywhere else: the two arithmetics differ by fp32 rounding of sums of ~1e3 (B), far below anything an alignment can see
    assert max(float(np.abs(x - y).max()) for x, y in zip(ll, ll_b)) < 2e-3
    assert all(np.isfinite(x).all() for x in ll_b)
    # same per-Gaussian fmaf chains; the log-sum-exp folds the Gaussians in a different order: <= 2 float ulps
    worst = max(float(np.max(np.abs(x - y) / np.spacing(np.abs(x)))) for x, y in zip(ll, ll_utt))
    same = sum(int((x == y).sum()) for x, y in zip(ll, ll_utt)) / sum(x.size for x in ll)
    assert worst <= 2.0 and same > 0.5, (worst, same)
    assert all(np.isfinite(x).all() for x in ll)

    # ---- K2 ----
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=200.0, acoustic_scale=0.1)
    assert not np.any(res["status"] & 1)
    poff, pdfs = us.pdf_lists()
    g = ut.graphs
    scale = np.float32(0.1)

    def path_cost(u, ali):
        """Walk the graph along `ali`; -> (is accepting path, cost in the token arithmetic of faster-decoder.h:119-137)."""
        s0 = int(g["state_off"][u]); ao = g["arc_off"]
        col = {int(p): j for j, p in enumerate(pdfs[poff[u]: poff[u + 1]])}
        st = int(g["start"][u])
        terms = np.empty(2 * len(ali), np.float64)
        for t, tid in enumerate(ali):
            a0, a1 = int(ao[s0 + st]), int(ao[s0 + st + 1])
            k = np.nonzero(g["ilabel"][a0:a1] == tid)[0]
            if k.size != 1:
                return False, np.inf
            a = a0 + int(k[0])
            w = np.float32(g["weight"][a] + cost[tid])                       # AddTransitionProbs: float add
            ac = np.float32(-1) * (scale * ll_b[u][col[int(m.id2pdf[tid])], t])  # decodable-am-diag-gmm.h:96
            terms[2 * t], terms[2 * t + 1] = w, ac
            st = int(g["nextstate"][a])
        fin = g["final"][s0 + st]
        if not np.isfinite(fin):
            return False, np.inf
        return True, float(np.add.accumulate(terms)[-1] + np.float64(fin))    # sequential (prev + w) + ac

    # the oracle's FasterDecoder + acc-stats on the first 240 utterances of this very set: identical alignments, statistics to 2e-5
    from helpers import assert_matches_oracle_replay, oracle_replay
    keep = oracle_replay(m, gc, ut, cost, 240, acoustic_scale=0.1, beam=200.0)
    assert_matches_oracle_replay(ctx, dm, tm, ut, res, keep, D)

    rng = np.random.default_rng(0)
    for u in rng.choice(U, size=120, replace=False):
        a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
        ok, c = path_cost(u, a)
        assert ok, f"utterance {u}: not an accepting path"
        assert -c / 0.1 == pytest.approx(float(res["like"][u]), rel=2e-6)      # decoder-wrappers.cc:93-95: / acoustic_scale
        ok_ref, c_ref = path_cost(u, ut.ref_ali[ut.frame_off[u]: ut.frame_off[u + 1]])
        assert ok_ref and c <= c_ref + 1e-9

    # ---- K3 ----
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs)
    got = accs.download()
    assert got["total_frames"] == N
    assert got["occ"].sum() == pytest.approx(N, rel=1e-6)
    assert np.array_equal(got["trans_acc"], np.bincount(res["ali"], minlength=m.num_tids + 1).astype(np.float64))
    x = ut.feats.astype(np.float64)
    np.testing.assert_allclose(got["mean_acc"].sum(0), x.sum(0), rtol=1e-5, atol=1e-6 * np.abs(x).sum(0).max())
    np.testing.assert_allclose(got["var_acc"].sum(0), (x * x).sum(0), rtol=1e-5)
    # reproducibility: one block per pdf at this size, buckets in frame order, waves folded in order -> a second
    # pass gives the same bits (total_log_like, a diagnostic summed across blocks with atomics, is excluded)
    accs2 = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs2)
    again = accs2.download()
    for k in ("occ", "mean_acc", "var_acc", "trans_acc"):
        assert np.array_equal(got[k], again[k]), k
    # per pdf: occupancy of a pdf's Gaussians = number of frames aligned to it
    per_pdf = np.add.reduceat(got["occ"], m.gauss_off[:-1].astype(np.int64))
    frames_pdf = np.bincount(m.id2pdf[res["ali"]], minlength=P)
    np.testing.assert_allclose(per_pdf, frames_pdf, rtol=1e-5, atol=1e-4)
