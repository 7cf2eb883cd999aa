"""The alignment half of the oracle has no known-answer test in the reference ("parity unpinned",
SURVEY.md 8c).  What can be checked on the CPU: the line-faithful FasterDecoder restatement agrees
with an independent exact Viterbi whenever its beam cannot bite, returns a cost >= the exact optimum
otherwise, and reproduces the status / retry / error semantics of AlignUtteranceWrapper."""
import numpy as np
import pytest

from graphs import hub_graph, random_graph
from oracle import oracle as orc


def _path_cost(g, id2pdf, ali_tids, ll, pdfs, scale):
    """Best cost over all paths whose emitting labels equal ali_tids (tiny DP), to score an alignment."""
    col = {p: j for j, p in enumerate(pdfs)}
    S = len(g["final"])
    src = np.repeat(np.arange(S), np.diff(g["arc_off"]))
    cur = np.full(S, np.inf); cur[g["start"]] = 0.0

    def eps(c):
        for _ in range(S + 1):
            for a in range(len(src)):
                if g["ilabel"][a] == 0 and c[src[a]] + g["weight"][a] < c[g["nextstate"][a]]:
                    c[g["nextstate"][a]] = c[src[a]] + g["weight"][a]
        return c
    cur = eps(cur)
    for t, tid in enumerate(ali_tids):
        nxt = np.full(S, np.inf)
        for a in range(len(src)):
            if g["ilabel"][a] == tid and cur[src[a]] < np.inf:
                ac = np.float32(-1) * (np.float32(scale) * ll[col[id2pdf[tid]], t])
                v = (cur[src[a]] + g["weight"][a]) + float(ac)
                nxt[g["nextstate"][a]] = min(nxt[g["nextstate"][a]], v)
        cur = eps(nxt)
    return float((cur + g["final"]).min())


@pytest.mark.parametrize("seed", range(12))
def test_faster_decoder_equals_exact_viterbi_when_beam_is_wide(seed):
    rng = np.random.default_rng(seed)
    num_tids = 12
    id2pdf = np.concatenate([[0], rng.integers(0, 6, size=num_tids)]).astype(np.int32)
    g = random_graph(rng, num_tids, n_main=int(rng.integers(3, 10)))
    T = int(rng.integers(len(g["final"]), 40))
    pdfs = np.arange(6, dtype=np.int32)
    ll = (-5 * rng.random((6, T)) - 1).astype(np.float32)
    og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
    ex = orc.exact_viterbi_ll(og, id2pdf, T, pdfs, ll, acoustic_scale=0.7)
    fd = orc.align_utterance_ll(og, id2pdf, T, pdfs, ll, acoustic_scale=0.7, beam=1e4)
    assert ex["status"] == 0 and fd["status"] == 0
    cost_fd = _path_cost(g, id2pdf, fd["ali"], ll, pdfs, 0.7)
    assert cost_fd == pytest.approx(ex["cost"], rel=1e-12)
    assert fd["like"] == pytest.approx(-ex["cost"] / 0.7, rel=1e-5)
    # narrow beam: still a valid path, never better than the optimum
    nb = orc.align_utterance_ll(og, id2pdf, T, pdfs, ll, acoustic_scale=0.7, beam=0.5, retry_beam=2.0)
    if (nb["status"] & 1) == 0:
        assert _path_cost(g, id2pdf, nb["ali"], ll, pdfs, 0.7) >= ex["cost"] - 1e-9
        assert len(nb["ali"]) == T


@pytest.mark.parametrize("seed", range(6))
def test_faster_decoder_on_hub_graphs_with_epsilon_ties(seed):
    """The graphs of tests/test_gpu_api.py::test_wave_faithful_decoder_epsilon_arcs_and_wide_fanout_...: a start state fanning out
    to 9-23 branches, epsilon chains and equal-cost epsilon pairs.  With a beam that cannot bite the oracle's FasterDecoder must
    find the exact optimum of an independent Viterbi (tiny DP with an epsilon closure) -- ties may pick a different path, never a
    different cost -- and with narrow beams never a better one."""
    rng = np.random.default_rng(100 + seed)
    num_tids = 20
    id2pdf = np.concatenate([[0], rng.integers(0, 8, size=num_tids)]).astype(np.int32)
    g = hub_graph(rng, num_tids, fan=int(rng.integers(9, 24)), tail=int(rng.integers(3, 9)), eps_ties=bool(seed % 2))
    T = int(rng.integers(12, 40))
    pdfs = np.arange(8, dtype=np.int32)
    ll = (-0.25 * rng.integers(0, 24, size=(8, T))).astype(np.float32)      # coarse grid: exact cost ties between paths occur
    og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
    ex = orc.exact_viterbi_ll(og, id2pdf, T, pdfs, ll, acoustic_scale=1.0)
    fd = orc.align_utterance_ll(og, id2pdf, T, pdfs, ll, acoustic_scale=1.0, beam=1e4)
    assert ex["status"] == 0 and fd["status"] == 0 and len(fd["ali"]) == T
    assert _path_cost(g, id2pdf, fd["ali"], ll, pdfs, 1.0) == pytest.approx(ex["cost"], rel=1e-12)
    for kw in (dict(beam=1.5, retry_beam=6.0), dict(beam=3.0, max_active=12, min_active=3), dict(beam=2.0, retry_beam=8.0, min_active=0)):
        nb = orc.align_utterance_ll(og, id2pdf, T, pdfs, ll, acoustic_scale=1.0, **kw)
        if (nb["status"] & 1) == 0:
            assert len(nb["ali"]) == T
            assert _path_cost(g, id2pdf, nb["ali"], ll, pdfs, 1.0) >= ex["cost"] - 1e-9


def test_status_semantics():
    rng = np.random.default_rng(0)
    id2pdf = np.array([0, 0, 1, 2], np.int32)
    pdfs = np.arange(3, dtype=np.int32)
    g = random_graph(rng, 3, n_main=6, p_eps=0.0, p_branch=0.0)
    og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
    ll = (-rng.random((3, 4))).astype(np.float32)
    # too few frames to reach the final state: error, and retried when a retry beam is given
    r = orc.align_utterance_ll(og, id2pdf, 4, pdfs, ll, beam=10.0)
    assert r["status"] == 1 and len(r["ali"]) == 0
    r = orc.align_utterance_ll(og, id2pdf, 4, pdfs, ll, beam=10.0, retry_beam=40.0)
    assert r["status"] == 3
    # decoder-wrappers.cc:29-33: retry_beam <= beam or beam <= 0 throws
    with pytest.raises(orc.OracleError):
        orc.align_utterance_ll(og, id2pdf, 4, pdfs, ll, beam=10.0, retry_beam=5.0)
    with pytest.raises(orc.OracleError):
        orc.align_utterance_ll(og, id2pdf, 4, pdfs, ll, beam=0.0)
    # empty graph: error (decoder-wrappers.cc:35-41)
    empty = orc.OGraph(-1, np.zeros(1, np.int32), [], [], [], [], [])
    assert orc.align_utterance_ll(empty, id2pdf, 4, pdfs, ll)["status"] == 1
    # words come from olabels along the path
    g2 = random_graph(np.random.default_rng(3), 3, n_main=4, p_eps=1.0, p_branch=0.0)
    og2 = orc.OGraph(g2["start"], g2["arc_off"], g2["ilabel"], g2["olabel"], g2["weight"], g2["nextstate"], g2["final"])
    ll2 = (-np.random.default_rng(4).random((3, 12))).astype(np.float32)
    r2 = orc.align_utterance_ll(og2, id2pdf, 12, pdfs, ll2)
    assert r2["status"] == 0 and len(r2["ali"]) == 12 and all(w > 0 for w in r2["words"])


def test_careful_graph_structure():
    # decoder-wrappers.cc:111-140: 2S+1 states, finals moved to the pre-initial state of the copy
    g = random_graph(np.random.default_rng(1), 4, n_main=3, p_eps=0.0)
    og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
    cg = orc.careful_graph(og)
    S = len(g["final"])
    assert cg.final.shape[0] == 2 * S + 1
    assert np.isinf(cg.final[: 2 * S]).all() and cg.final[2 * S] == 0.0
    assert cg.ilabel.shape[0] == 2 * g["ilabel"].shape[0] + 2


def test_gmm_decodable_path_matches_matrix_path():
    from kaldi_hmm_gmm_amd import synth
    m = synth.make_model(12, 3, 6, seed=4)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    om = orc.OModel(m.gauss_off, gc, m.means_invvars, m.inv_vars)
    ut = synth.make_utts(m, 3, seed=9, min_phones=2, max_phones=3)
    for u in range(3):
        g = orc.OGraph.from_set(ut.graphs, u)
        f = ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]]
        a = orc.align_utterance(g, om, m.id2pdf, f, acoustic_scale=0.1)
        pdfs = np.arange(12, dtype=np.int32)
        ll = orc.loglikes_matrix(om, f, pdfs)
        b = orc.align_utterance_ll(g, m.id2pdf, f.shape[0], pdfs, ll, acoustic_scale=0.1)
        assert a["status"] == b["status"] == 0 and (a["ali"] == b["ali"]).all() and a["like"] == b["like"]
        assert a["loglike_evals"] <= f.shape[0] * 12


def test_threaded_pass_keeps_what_the_one_thread_path_computes():
    """orc_em_pass_mt_keep (the utterance-parallel driver bench.py times, keeping its results for the parity check at scale): same
    alignments, status and like as orc_align_utterance per utterance, and accumulators that sum to the one-thread accumulator."""
    import numpy as np

    from helpers import build, utt_feats
    from oracle import oracle as orc

    m, gc, om, ut, cost = build(30, 8, 20, n_utt=12, seed=5)
    g = dict(ut.graphs)
    g["weight"] = np.where(g["ilabel"] >= 1, g["weight"] + cost[g["ilabel"]], g["weight"]).astype(np.float32)
    keep = {}
    fr, nn, failed, _ = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, first_utt=2, n_utt=9, num_threads=3, acoustic_scale=0.1,
                                       keep=keep)
    assert nn == 9 and failed == 0 and fr == ut.frame_off[11] - ut.frame_off[2] and (keep["status"] == 0).all()
    oa = orc.OAccs(int(m.gauss_off[-1]), m.dim, m.num_tids)
    for u in range(2, 11):
        r = orc.align_utterance(orc.OGraph.from_set(g, u), om, m.id2pdf, utt_feats(ut, u), acoustic_scale=0.1)
        a = keep["ali"][ut.frame_off[u] - ut.frame_off[2]: ut.frame_off[u + 1] - ut.frame_off[2]]
        assert (a == r["ali"]).all() and keep["like"][u - 2] == np.float32(r["like"])
        orc.acc_stats_ali(om, m.id2pdf, utt_feats(ut, u), r["ali"], oa)
    np.testing.assert_allclose(keep["accs"].occ, oa.occ, rtol=1e-12)
    np.testing.assert_allclose(keep["accs"].mean_acc, oa.mean_acc, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(keep["accs"].var_acc, oa.var_acc, rtol=1e-10, atol=1e-12)
    assert (keep["accs"].trans_acc == oa.trans_acc).all() and keep["accs"].total_frames == oa.total_frames
    # a budget of zero seconds reaches nothing: status stays -1
    keep = {}
    fr, nn, failed, _ = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, first_utt=0, n_utt=12, num_threads=2, budget_seconds=0.0,
                                       acoustic_scale=0.1, keep=keep)
    assert nn == 0 and (keep["status"] == -1).all() and keep["accs"].occ.sum() == 0
