"""One EM pass at every single-GPU configuration BASELINE.json names, as it names them -- #2 mono 100 pdfs x 8 Gaussians, D = 39, 1000
utterances; #3 2000 x 32, D = 40, 10 000 utterances (3 M frames) -- and at the benchmark's model shape (#4: 5000 x 64 x 40,
bench-like utterances of 10..40 phones; the full 100 000 utterances are bench.py's `check`, on the driver's record).
The oracle replays the first 240 utterances (~72 k frames: a few seconds on the host's cores, orc_em_pass_mt_keep) and K2 / K3 must
give its alignments / statistics on them; the whole set goes through size-independent properties:

  K1  the pdf-major and the utterance-major kernels (two independent tilings) agree within 2 float ulps on every cell
  K2  every alignment is an accepting path of its graph; the returned likelihood is that path's cost replayed on
      the host in the reference's token arithmetic; the path is never worse than the generating path (no pruning)
  K3  sum(occ) = frames, transition counts = histogram of the alignment, sum_g mean_acc = sum_t x_t and
      sum_g var_acc = sum_t x_t^2 (posteriors of a frame sum to one)

test_hard_model_at_recipe_beams: the same shape with a MISMATCHED model (synth.mismatched_model) at the recipe's beams (6, retry 40)
and at beam 1: the regime where the reference's pruning decides the answer -- the oracle's replay there holds >= 100 utterances
that went through the order-faithful decoder (csrc/faster-decoder.cc:154-335) and, at beam 1, as many retried ones
(csrc/decoder-wrappers.cc:55-77).
"""
import numpy as np
import pytest

from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,G,D,U,seed", [(100, 8, 39, 1000, 20230415), (2000, 32, 40, 10000, 20230416), (5000, 64, 40, 1500, 20230417)],
                         ids=["cfg2_mono100x8_1k_utts", "cfg3_tri2000x32_10k_utts", "cfg4_shape_tri5000x64_1500_utts"])
def test_em_pass_properties_at_bench_shape(ctx, opt, P, G, D, U, seed):
    m = synth.make_model(P, G, D, seed=seed)
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

    # ---- K1: two fp32-MFMA tilings, one answer; the f16x2s form (the default) within the fp32 bound of them ----
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
    accs.close(); accs2.close(); us.close(); tm.close(); dm.close()


@pytest.mark.parametrize("beam,retry,min_active", [(6.0, 40.0, 20), (0.5, 40.0, 0)])
def test_hard_model_at_recipe_beams(ctx, opt, beam, retry, min_active):
    """5000 x 64 x 40 with a MISMATCHED model (a tenth of the pdfs traded parameters: sure of itself and wrong there -- a recipe's
    early realign passes in caricature) at the recipe's beams 6 / retry 40 (egs/yesno/train.py:165-168) and at beam 0.5 / retry 40 with
    FasterDecoderOptions.min_active = 0 (python/csrc/faster-decoder.cc:14-53; with the default 20 a chain graph never loses its last tokens):
    the best path leaves the beam, the certificate fails and the order-faithful decoder -- FasterDecoder's ProcessEmitting /
    GetCutoff (csrc/faster-decoder.cc:154-335) under AlignUtteranceWrapper's retry (csrc/decoder-wrappers.cc:55-77) -- produces the
    answer.  The oracle replays the first 300 utterances; >= 100 of them went through that decoder here (at beam 1 nearly all of
    them after a failed first attempt), and every alignment, status (done / retried / error) and likelihood equals the oracle's;
    K3 on the oracle's alignments gives its statistics (failed utterances contribute nothing)."""
    from helpers import assert_matches_oracle_replay, oracle_replay
    P, G, D, U, NR = 5000, 64, 40, 1500, 300
    m0 = synth.make_model(P, G, D, seed=20230418)
    ut = synth.make_utts(m0, U, seed=6)
    m = synth.mismatched_model(m0, 0.1, seed=2)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    il = np.arange(m.num_tids + 1, dtype=np.int32)
    cost = orc.add_transition_probs(il, np.zeros(m.num_tids + 1, np.float32), m.log_probs, m.non_self_loop_log_probs,
                                    m.id2state, m.is_self_loop, 1.0, 0.1)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1, min_active=min_active)
    st = res["status"]
    n_fb, n_retry, n_err = int((st[:NR] & 8 != 0).sum()), int((st[:NR] & 2 != 0).sum()), int((st[:NR] & 1).sum())
    print(f"beam {beam} / {retry}, replayed {NR}: {n_fb} through the order-faithful decoder, {n_retry} retried, {n_err} failed; whole set: "
          f"{int((st & 8 != 0).sum())} / {int((st & 2 != 0).sum())} / {int((st & 1).sum())} of {U}")
    assert n_fb >= 100, n_fb
    if beam < 2.0:
        assert n_retry >= 3, n_retry
    # the oracle decodes its OWN fp32 scores (csrc/decodable-am-diag-gmm.cc:55-61); K1's differ by ~1e-7 B: identical answers all the same
    keep = oracle_replay(m, gc, ut, cost, NR, acoustic_scale=0.1, beam=beam, retry_beam=retry, min_active=min_active)
    assert int((keep["status"] & 2 != 0).sum()) == n_retry and int((keep["status"] & 1).sum()) == n_err
    # (statistics: a frame under a traded pdf has B ~ 1e3 -- fp32 posteriors are defined to ~1e-3 there, see the helper; the device
    #  must also be no farther from a float64 accumulation than the oracle's fp32 chain is)
    assert_matches_oracle_replay(ctx, dm, tm, ut, res, keep, D, stats_rtol=1e-3, exact=(m, gc))
    # size-independent properties over the whole set: failed utterances carry an all-zero alignment, the others none
    for u in range(U):
        a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
        assert (a == 0).all() if st[u] & 1 else (a > 0).all(), u
    # The asynchronous call pattern of an EM pass (bench.py's step: khg_align without host outputs, then khg_acc_stats): the SPLIT mode --
    # the order-faithful decoders write to their own buffer on a side stream while K3 accumulates the certified utterances, the rest
    # follows in a second pass (statistics are additive, csrc/mle-am-diag-gmm.cc:41-52) -- against the synchronous path above:
    # the same alignment, the same transition counts, sums to the fp16-split phase B's grouping tolerance (32-frame fp32 partial
    # sums: which frames of a pdf share a group differs between one pass and two).
    acc_sync = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, acc_sync)
    want = acc_sync.download()
    for split_opt in (0, 1):
        opt("k2_split", split_opt)
        us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1, min_active=min_active, download=False)
        acc_async = DeviceAccs(ctx, dm, tm)
        us.acc_stats(dm, tm, acc_async)
        got = acc_async.download()
        assert np.array_equal(np.asarray(us.download_ali()), res["ali"]), split_opt
        assert np.array_equal(got["trans_acc"], want["trans_acc"]) and got["total_frames"] == want["total_frames"], split_opt
        for k in ("occ", "mean_acc", "var_acc"):
            np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-6 * np.abs(want[k]).max(), err_msg=f"{k} split_opt {split_opt}")
        assert got["total_log_like"] == pytest.approx(want["total_log_like"], rel=1e-9)
        acc_async.close()
    acc_sync.close()
    us.close(); tm.close(); dm.close()
