"""Shared helpers for the parity tests: build oracle / device objects from a synthetic set."""
import numpy as np

from kaldi_hmm_gmm_amd import synth
from oracle import oracle as orc


def build(num_pdfs, gauss, dim, n_utt, seed=1, ragged=False, min_phones=2, max_phones=6, tscale=1.0, slscale=0.1, transcripts="uniform", gauss_counts=None):
    m = synth.make_model(num_pdfs, gauss, dim, seed=20230414 + seed, ragged=ragged, gauss_counts=gauss_counts)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    om = orc.OModel(m.gauss_off, gc, m.means_invvars, m.inv_vars)
    ut = synth.make_utts(m, n_utt, seed=seed, min_phones=min_phones, max_phones=max_phones, transcripts=transcripts)
    # what AddTransitionProbs adds per tid, from the oracle (hmm-utils.cc:442-463)
    il = np.arange(m.num_tids + 1, dtype=np.int32)
    cost = orc.add_transition_probs(il, np.zeros(m.num_tids + 1, np.float32), m.log_probs, m.non_self_loop_log_probs,
                                    m.id2state, m.is_self_loop, tscale, slscale)
    return m, gc, om, ut, cost


def oracle_graph(ut, u, cost):
    g = dict(ut.graphs)
    w = g["weight"].copy()
    il = g["ilabel"]
    w = np.where(il >= 1, w + cost[np.maximum(il, 0)], w).astype(np.float32)
    g["weight"] = w
    return orc.OGraph.from_set(g, u)


def utt_feats(ut, u):
    return ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]]


def exact_loglikes(m, gc, feats, pdfs):
    """float64 evaluation of decodable-am-diag-gmm.cc:55-61 and the magnitude bound
    B = max_g(|gconst| + sum|M x| + 0.5 sum|V x^2|) that fp32 rounding error scales with."""
    x = feats.astype(np.float64)
    out = np.zeros((len(pdfs), x.shape[0]))
    bound = np.zeros_like(out)
    for j, p in enumerate(pdfs):
        a, b = m.gauss_off[p], m.gauss_off[p + 1]
        miv = m.means_invvars[a:b].astype(np.float64)
        iv = m.inv_vars[a:b].astype(np.float64)
        g = gc[a:b].astype(np.float64)
        ll = g[None, :] + x @ miv.T - 0.5 * (x * x) @ iv.T
        mx = ll.max(1, keepdims=True)
        out[j] = (mx + np.log(np.exp(ll - mx).sum(1, keepdims=True)))[:, 0]
        bb = np.abs(g)[None, :] + np.abs(x) @ np.abs(miv).T + 0.5 * (x * x) @ iv.T
        bound[j] = bb.max(1)
    return out, bound


def token_path_cost(graphs, u, ali, ll_u, pdf_list, cost, id2pdf, acoustic_scale):
    """Walk utterance u's graph along the transition-id sequence `ali` (out-arcs of a state carry distinct ids in the
    compiled training graphs): -> (is an accepting path, its cost in the token arithmetic of faster-decoder.h:119-137:
    double(prev) + float(arc weight + AddTransitionProbs) + float(-(scale * ll)), left to right, + final weight)."""
    g = graphs
    s0 = int(g["state_off"][u])
    ao = g["arc_off"]
    col = {int(p): j for j, p in enumerate(pdf_list)}
    st = int(g["start"][u])
    scale = np.float32(acoustic_scale)
    terms = np.empty(2 * len(ali), np.float64)
    for t, tid in enumerate(ali):
        a0, a1 = int(ao[s0 + st]), int(ao[s0 + st + 1])
        k = np.nonzero(g["ilabel"][a0:a1] == tid)[0]
        if k.size != 1:
            return False, np.inf
        a = a0 + int(k[0])
        terms[2 * t] = np.float32(g["weight"][a] + cost[tid])                              # AddTransitionProbs: float add
        terms[2 * t + 1] = np.float32(-1) * (scale * ll_u[col[int(id2pdf[tid])], t])        # decodable-am-diag-gmm.h:96
        st = int(g["nextstate"][a])
    fin = g["final"][s0 + st]
    if not np.isfinite(fin):
        return False, np.inf
    return True, float(np.add.accumulate(terms)[-1] + np.float64(fin))


def oracle_replay(m, gc, ut, cost, n_utt, acoustic_scale=0.1, beam=200.0, retry_beam=0.0, threads=None, **kw):
    """The oracle's own answer for the first `n_utt` utterances of a set, computed utterance-parallel inside the C oracle
    (orc_em_pass_mt_keep: orc_align_utterance + orc_acc_stats_ali per utterance, the calls the one-thread path makes): alignments,
    status, like and the accumulators of those utterances.  ~30 k frames/s per core at 5000 x 64 x 40."""
    import os

    om = orc.OModel(m.gauss_off, gc, m.means_invvars, m.inv_vars)
    g = dict(ut.graphs)
    g["weight"] = np.where(g["ilabel"] >= 1, g["weight"] + cost[np.maximum(g["ilabel"], 0)], g["weight"]).astype(np.float32)
    keep = {}
    nthr = threads or max(1, min(len(os.sched_getaffinity(0)), 16))
    fr, nn, failed, _ = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, first_utt=0, n_utt=n_utt, num_threads=nthr,
                                       acoustic_scale=acoustic_scale, beam=beam, retry_beam=retry_beam, keep=keep, **kw)
    assert nn == n_utt
    keep["n_utt"], keep["frames"] = n_utt, int(ut.frame_off[n_utt])
    return keep


def assert_matches_oracle_replay(ctx, dm, tm, ut, res, keep, dim, stats_rtol=2e-5, exact=None):
    """K2's result `res` (ali / status / like of the whole set) against the oracle's on the replayed utterances: identical alignments,
    same status bits, like to 2e-5; then K3 over exactly those utterances (from the oracle's alignment) against the oracle's
    accumulators at the tolerances of tests/test_gpu_parity.py.
    stats_rtol: the statistics' relative tolerance.  2e-5 for a model that matches its data.  Posteriors are exp() of log-likelihood
    differences, and an fp32 log-likelihood carries ~1e-6 B (B = |gconst| + sum|M x| + 0.5 sum V x^2, helpers.exact_loglikes): a frame
    under a pdf that does not match it has B ~ 1e3, so the reference's own statistics are only defined to ~1e-3 there.
    exact = (model, gconsts): additionally accumulate the occupancies in float64 on the host and require the device's distance to
    them to be no larger than the oracle's own (x 2, + 1e-9): the looser tolerance is fp32's, not this library's."""
    import pytest

    from kaldi_hmm_gmm_amd import DeviceAccs, UtteranceSet

    n, nfr = keep["n_utt"], keep["frames"]
    assert ((res["status"][:n] & 3) == (keep["status"] & 3)).all()
    ok = (keep["status"] & 1) == 0
    bad = [u for u in range(n) if not np.array_equal(res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]], keep["ali"][ut.frame_off[u]: ut.frame_off[u + 1]])]
    assert not bad, f"{len(bad)} of {n} utterances align differently from the oracle (first: {bad[:5]})"
    assert res["like"][:n][ok] == pytest.approx(keep["like"][ok], rel=2e-5)
    sub = UtteranceSet(ctx, None, ut.frame_off[: n + 1].astype(np.int64), np.ascontiguousarray(ut.feats[:nfr]))
    sub.upload_ali(np.ascontiguousarray(keep["ali"][:nfr], np.int32))
    accs = DeviceAccs(ctx, dm, tm)
    sub.acc_stats(dm, tm, accs)
    got = accs.download()
    oa = keep["accs"]
    assert np.array_equal(got["trans_acc"], oa.trans_acc) and got["total_frames"] == oa.total_frames
    np.testing.assert_allclose(got["occ"], oa.occ, rtol=stats_rtol, atol=1e-6)
    np.testing.assert_allclose(got["mean_acc"], oa.mean_acc, rtol=stats_rtol, atol=2e-6 * np.abs(oa.mean_acc).max())
    np.testing.assert_allclose(got["var_acc"], oa.var_acc, rtol=stats_rtol, atol=2e-6 * np.abs(oa.var_acc).max())
    assert got["total_log_like"] == pytest.approx(oa.total_log_like, rel=1e-5)
    sub.close(); accs.close()
    if exact is not None:
        m, gc = exact
        ali = keep["ali"][:nfr]
        pdf = np.where(ali > 0, m.id2pdf[np.maximum(ali, 0)], -1)
        occ64 = np.zeros_like(oa.occ)
        x64 = ut.feats[:nfr].astype(np.float64)
        order = np.argsort(pdf, kind="stable")
        bounds = np.searchsorted(pdf[order], np.arange(m.num_pdfs + 1))
        for p in np.unique(pdf[pdf >= 0]):
            fr = order[bounds[p]: bounds[p + 1]]
            a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
            ll = gc[a:b].astype(np.float64)[None, :] + x64[fr] @ m.means_invvars[a:b].astype(np.float64).T \
                - 0.5 * (x64[fr] ** 2) @ m.inv_vars[a:b].astype(np.float64).T
            e = np.exp(ll - ll.max(1, keepdims=True))
            occ64[a:b] = (e / e.sum(1, keepdims=True)).sum(0)
        err_dev, err_orc = np.abs(got["occ"] - occ64).max(), np.abs(oa.occ - occ64).max()
        print(f"occupancies against float64: device max |err| {err_dev:.3g}, oracle (the reference's fp32 chain) {err_orc:.3g}")
        assert err_dev <= 2.0 * err_orc + 1e-9, (err_dev, err_orc)
