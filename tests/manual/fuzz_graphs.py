"""Randomised decoder sweep over GENERAL graphs (not part of pytest): left-to-right graphs with branches, epsilon-input arcs
carrying word labels, unreachable finals, empty graphs, utterances too short to reach the end; random beams (pruning,
retries, max_active).  K1's own scores feed both sides: status / alignment / words must match the oracle exactly.
python tests/manual/fuzz_graphs.py [seconds] [seed]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
from graphs import concat, random_graph
from oracle import oracle as orc
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
t0 = time.time(); n = nutt = nfall = nerr = nret = 0
EMPTY = {"start": -1, "arc_off": np.zeros(1, np.int64), "ilabel": np.zeros(0, np.int32), "olabel": np.zeros(0, np.int32),
         "weight": np.zeros(0, np.float32), "nextstate": np.zeros(0, np.int32), "final": np.zeros(0, np.float32)}
while time.time() - t0 < budget:
    P = int(rng.choice([3, 6, 12, 30])); G = int(rng.choice([1, 3, 8])); D = int(rng.choice([2, 8, 13]))
    seed = int(rng.integers(1 << 30))
    m = synth.make_model(P, G, D, seed=seed)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    U = int(rng.integers(1, 14))
    p_eps = float(rng.choice([0.0, 0.0, 0.2, 0.6])); p_br = float(rng.choice([0.0, 0.3, 0.8])); p_long = float(rng.choice([0.0, 0.5, 0.9]))
    graphs = []
    for _ in range(U):
        r = rng.random()
        if r < 0.04:
            graphs.append(EMPTY)
        else:
            graphs.append(random_graph(rng, m.num_tids, n_main=int(rng.integers(1, 40)), p_branch=p_br, p_eps=p_eps, with_final=r > 0.1, p_long=p_long))
    if all(g is EMPTY for g in graphs):       # a set without any state is a features-only set by the C-ABI's contract
        graphs[0] = random_graph(rng, m.num_tids, n_main=3)
    T = [int(rng.integers(max(1, len(g["final"]) - 2), len(g["final"]) + 40)) for g in graphs]
    if rng.random() < 0.2:
        T[int(rng.integers(U))] = 0
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    scale = float(rng.choice([0.1, 0.3, 1.0]))
    feats = (rng.standard_normal((max(int(frame_off[-1]), 1), D)) * float(rng.choice([0.5, 3.0]))).astype(np.float32)[: int(frame_off[-1])]
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(np.zeros(m.num_tids + 1, np.float32))
    us = UtteranceSet(ctx, tm, frame_off, feats if feats.shape[0] else np.zeros((0, D), np.float32), graphs=concat(graphs))
    us.loglikes(dm, reachable_only=bool(rng.integers(2)))
    us.loglikes(dm)                      # full scores for the oracle (unreadable cells are unspecified otherwise)
    lls = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    beam, retry = [(200.0, 0.0), (16.0, 0.0), (6.0, 40.0), (2.0, 8.0), (0.5, 1.0)][int(rng.integers(5))]
    kw = {}
    if rng.random() < 0.25:
        kw = {"max_active": int(rng.choice([2, 5, 30])), "min_active": int(rng.choice([0, 1]))}
    tag = f"P{P} G{G} D{D} U{U} eps{p_eps} br{p_br} long{p_long} beam{beam}/{retry} scale{scale} {kw} seed{seed}"
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=scale, **kw)
    for u, g in enumerate(graphs):
        og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
        ll = lls[u] if lls[u].size else np.zeros((1, max(T[u], 1)), np.float32)
        want = orc.align_utterance_ll(og, m.id2pdf, T[u], pdfs[poff[u]: poff[u + 1]], ll, acoustic_scale=scale, beam=beam, retry_beam=retry, **kw)
        st = int(res["status"][u])
        assert (st & 1) == (want["status"] & 1), (tag, u, st, want["status"])
        if g["start"] >= 0:
            assert (st & 2) == (want["status"] & 2), (tag, u, st, want["status"])
        a = res["ali"][frame_off[u]: frame_off[u + 1]]
        nfall += (st & 8) != 0; nret += (st & 2) != 0
        if want["status"] & 1:
            nerr += 1
            assert (a == 0).all(), (tag, u)
        else:
            assert (a == want["ali"]).all(), (tag, u)
            w = res["words"][res["words_off"][u]: res["words_off"][u + 1]]
            assert (w == want["words"]).all(), (tag, u, "words")
            assert abs(res["like"][u] - want["like"]) <= 1e-5 * abs(want["like"]) + 1e-4, (tag, u, "like")
    n += 1; nutt += U
    us.close(); tm.close(); dm.close()
print(f"graph fuzz ok: {n} batches, {nutt} utterances ({nerr} failed like the oracle, {nret} retried, {nfall} through the fallback decoders) in {time.time() - t0:.0f}s")
