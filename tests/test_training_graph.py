"""Training-graph builder + equal-align: structural contract of
scripts/test_training_graph_compiler.py:85-105 (a valid path of exactly T transition-ids), the
stochasticity of the compiled graph (path cost == -sum log P of its transitions and lexicon choices at
scales 1.0), and -- on the GPU -- alignment of the compiled graphs against the oracle decoder."""
import math

import numpy as np
import pytest

from oracle import oracle as orc


def _setup(sil=True):
    from kaldi_hmm_gmm_amd.context_dep import monophone_context_dependency
    from kaldi_hmm_gmm_amd.training_graph import TrainingGraphCompiler, generate_hmm_topo
    from kaldi_hmm_gmm_amd.transition_model import TransitionModel

    topo = generate_hmm_topo([2, 3, 4], 1)                 # SIL=1, three real phones
    cd = monophone_context_dependency(topo.phones, topo.get_phone_to_num_pdf_classes())
    tm = TransitionModel(cd, topo)
    lex = {1: [(1.0, [2])], 2: [(1.0, [3])], 3: [(0.7, [2, 4]), (0.3, [4])]}   # word 3 has two pronunciations
    gc = TrainingGraphCompiler(tm, cd, lex, sil_phone=1 if sil else None)
    return topo, cd, tm, gc


def _walk(g, ali):
    """All (state, cost) reachable by consuming `ali`; returns the cheapest accepting cost or None."""
    cur = {g.start: 0.0}
    for t in ali:
        nxt = {}
        for s, c in cur.items():
            for a in g.arcs(s):
                if a.ilabel == t:
                    v = c + a.weight
                    if v < nxt.get(a.nextstate, math.inf):
                        nxt[a.nextstate] = v
        cur = nxt
        if not cur:
            return None
    best = None
    for s, c in cur.items():
        if g.is_final(s):
            v = c + g.final(s)
            best = v if best is None or v < best else best
    return best


def test_topology_and_sizes():
    topo, cd, tm, gc = _setup()
    assert topo.get_phone_to_num_pdf_classes() == [-1, 5, 3, 3, 3]      # scripts/prepare_lang.py:514-600
    assert tm.num_pdfs == 5 + 9
    g = gc.compile_graph_from_text([1, 2])
    assert g.start == 0 and g.num_states > 0
    # epsilon-free, every state's incoming arcs share one transition-state (AddSelfLoopsReorder's precondition)
    cls = {}
    for s in range(g.num_states):
        for a in g.arcs(s):
            assert a.ilabel != 0
            if a.nextstate != s:
                ts = tm.transition_id_to_transition_state(a.ilabel)
                assert cls.setdefault(a.nextstate, ts) == ts
    # the self-loop on a state is the self-loop of the transition-state entering it ("reorder")
    for s in range(g.num_states):
        for a in g.arcs(s):
            if a.nextstate == s:
                assert a.ilabel == tm.self_loop_of(cls[s]) and a.olabel == 0


@pytest.mark.parametrize("sil", [True, False])
def test_equal_align_contract_and_stochasticity(sil):
    from kaldi_hmm_gmm_amd.training_graph import equal_align

    topo, cd, tm, gc = _setup(sil)
    logp = np.asarray(tm.log_probs, np.float64)
    for seed, words in enumerate([[1], [1, 2], [2, 3, 1], [3, 3]]):
        g = gc.compile_graph_from_text(words)
        for T in (40, 97):
            ok, ali = equal_align(g, T, rand_seed=seed + 3)
            assert ok and len(ali) == T                                    # scripts/test_training_graph_compiler.py:85-105
            cost = _walk(g, ali)
            assert cost is not None, "equal_align produced a sequence the graph does not accept"
            # path cost = -sum log P(transition) + lexicon costs: between (n+1)*log 2 ... with silence, pron costs for word 3
            trans = -logp[ali].sum()
            lexc = cost - trans
            nsil = sum(1 for t in ali if tm.transition_id_to_phone(t) == 1 and tm.transition_id_to_hmm_state(t) == 0
                       and not tm.is_self_loop(t))
            want = (len(words) + 1) * math.log(2.0) if sil else 0.0
            prons = [math.log(1 / 0.7), math.log(1 / 0.3)]
            n3 = words.count(3)
            cands = [want + sum(c) for c in ([()] if n3 == 0 else [(a,) for a in prons] if n3 == 1 else
                                             [(a, b) for a in prons for b in prons])]
            assert min(abs(lexc - c) for c in cands) < 1e-3, (words, lexc, cands, nsil)
        # too short: fewer frames than emitting states on the shortest path
        ok, _ = equal_align(g, 2)
        assert not ok


def test_words_and_min_length():
    topo, cd, tm, gc = _setup(True)
    g = gc.compile_graph_from_text([2, 1])
    c = g.to_csr()
    og = orc.OGraph(c["start"], c["arc_off"], c["ilabel"], c["olabel"], c["weight"], c["nextstate"], c["final"])
    # a flat model: every pdf scores 0 -> the cheapest path is decided by the graph alone (no silence: 6 states)
    id2pdf = np.asarray(tm.transition_id_to_pdf_array(), np.int32)
    T = 12
    npdf = tm.num_pdfs
    res = orc.align_utterance_ll(og, id2pdf, T, np.arange(npdf, dtype=np.int32), np.zeros((npdf, T), np.float32),
                                 acoustic_scale=1.0, beam=200.0, retry_beam=0.0)
    assert res["status"] == 0 and res["words"].tolist() == [2, 1]
    phones = [tm.transition_id_to_phone(int(t)) for t in res["ali"]]
    assert 1 not in phones, "optional silence costs log 2 more than skipping it on a flat model"
    res5 = orc.align_utterance_ll(og, id2pdf, 5, np.arange(npdf, dtype=np.int32), np.zeros((npdf, 5), np.float32),
                                  acoustic_scale=1.0, beam=200.0, retry_beam=0.0)
    assert res5["status"] & 1, "5 frames cannot cover 6 emitting states"


@pytest.mark.gpu
def test_compiled_graphs_align_like_oracle(ctx):
    """Graphs with the 5-state silence (in-degree 4: the 3-bit register-resident Viterbi path) through the
    reference-style entry points vs the oracle's FasterDecoder."""
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd import _gpu
    from kaldi_hmm_gmm_amd.training_graph import equal_align

    _gpu.set_default_context(ctx)
    topo, cd, tm, gc = _setup(True)
    rng = np.random.default_rng(11)
    D = 8
    allx = (rng.standard_normal((400, D)) * 2).astype(np.float32)
    tm2, tree, am = khg.gmm_init_mono(topo, allx)
    assert tm2.num_pdfs == tm.num_pdfs
    means = (rng.standard_normal((tm.num_pdfs, D)) * 3).astype(np.float32)
    for p in range(tm.num_pdfs):
        gm = am.get_pdf(p); gm.set_means(means[p][None, :]); gm.compute_gconsts()
    go, gcst, w, miv, iv = am.flat()
    om = orc.OModel(go, gcst, miv, iv)
    id2pdf = np.asarray(tm.transition_id_to_pdf_array(), np.int32)
    names, fsts, feats = [], [], []
    for u, words in enumerate([[1, 2], [3], [2, 3, 1], [1, 1, 2, 3]]):
        g = gc.compile_graph_from_text(words)
        T = 30 + 17 * u
        ok, ali = equal_align(g, T, rand_seed=u)
        assert ok
        x = np.stack([means[id2pdf[t]] for t in ali]).astype(np.float32) + rng.standard_normal((T, D)).astype(np.float32)
        names.append(f"u{u}"); fsts.append(g); feats.append(x)
    for beam, retry in ((200.0, 0.0), (6.0, 40.0)):
        cfg = khg.AlignConfig(beam=beam, retry_beam=retry)
        rb = khg.gmm_align_compiled_batch(am, tm, names, [f.copy() for f in fsts], feats, cfg, acoustic_scale=0.1,
                                          transition_scale=1.0, self_loop_scale=0.1)
        cost = tm.scaled_trans_cost(1.0, 0.1)
        for u, g in enumerate(fsts):
            c = g.to_csr()
            wgt = np.where(c["ilabel"] >= 1, c["weight"] + cost[np.maximum(c["ilabel"], 0)], c["weight"]).astype(np.float32)
            og = orc.OGraph(c["start"], c["arc_off"], c["ilabel"], c["olabel"], wgt, c["nextstate"], c["final"])
            want = orc.align_utterance(og, om, id2pdf, feats[u], acoustic_scale=0.1, beam=beam, retry_beam=retry)
            assert want["status"] & 1 == 0
            assert rb["alignment"][u] == want["ali"].tolist(), (beam, u)
            assert rb["words"][u] == want["words"].tolist()
        assert rb["num_done"] == len(fsts) and rb["num_error"] == 0


def test_word_loop_decoding_graph_accepts_what_training_graphs_accept():
    """compile_word_loop_graph (the decoding graph of examples/decode_synthetic.py): epsilon-free, the start
    state is final, and any transition-id sequence a TRAINING graph of some transcript accepts is accepted by the
    loop graph too, at the training graph's cost plus the unigram cost of its words (-log(1/3) each)."""
    topo, cd, tm, gc = _setup()
    loop = gc.compile_word_loop_graph()
    assert loop.start == 0 and loop.is_final(0)
    assert all(a.ilabel != 0 for s in range(loop.num_states) for a in loop.arcs(s))
    rng = np.random.default_rng(5)
    from kaldi_hmm_gmm_amd.training_graph import equal_align
    for transcript in ([1], [2, 1], [3, 3, 2], [1, 2, 3, 1]):
        g = gc.compile_graph_from_text(transcript)
        for seed in range(4):
            ok, ali = equal_align(g, 60 + 7 * seed, rand_seed=seed, num_retries=10)
            assert ok
            c_train, c_loop = _walk(g, ali), _walk(loop, ali)
            assert c_train is not None and c_loop is not None
            assert c_loop == pytest.approx(c_train + len(transcript) * math.log(3.0), abs=2e-4)
    # and the olabels along an accepted path spell the transcript: follow the cheapest path greedily
    g = gc.compile_graph_from_text([2, 3, 1])
    ok, ali = equal_align(g, 80, rand_seed=1, num_retries=10)
    cur = {loop.start: (0.0, [])}
    for t in ali:
        nxt = {}
        for s, (c, w) in cur.items():
            for a in loop.arcs(s):
                if a.ilabel == t and c + a.weight < nxt.get(a.nextstate, (math.inf,))[0]:
                    nxt[a.nextstate] = (c + a.weight, w + ([a.olabel] if a.olabel else []))
        cur = nxt
    finals = [(c + loop.final(s), w) for s, (c, w) in cur.items() if loop.is_final(s)]
    assert min(finals)[1] == [2, 3, 1]


@pytest.mark.parametrize("sil_disambig", [None, 9])
def test_lexicon_fst_form_equals_dict_form(sil_disambig):
    """TrainingGraphCompiler(trans_model=, ctx_dep=, lex_fst=, disambig_syms=, opts=) -- the reference's constructor
    (python/csrc/training-graph-compiler.cc:32-58, egs/yesno/train.py:70-76) with an L.fst built like
    scripts/prepare_lang.py:329-456 -- accepts the same transition-id sequences at the same costs as the dict form."""
    from kaldi_hmm_gmm_amd import TrainingGraphCompiler, TrainingGraphCompilerOptions, equal_align, make_lexicon_fst_with_silence

    topo, cd, tm, gc_dict = _setup(True)
    lex = {1: [(1.0, [2])], 2: [(1.0, [3])], 3: [(0.7, [2, 4]), (0.3, [4, 8])]}       # 8 = a disambiguation symbol (#1) on one pronunciation
    L = make_lexicon_fst_with_silence(lex, sil_phone=1, sil_prob=0.5, sil_disambig=sil_disambig)
    gc = TrainingGraphCompiler(trans_model=tm, ctx_dep=cd, lex_fst=L, disambig_syms=[7, 8, 9], opts=TrainingGraphCompilerOptions())
    assert L.num_states == gc.lex_fst.num_states            # the compiler works on its own copy
    for seed, words in enumerate([[1], [1, 2], [2, 3, 1], [3, 3], []]):
        a, b = gc_dict.compile_graph_from_text(words), gc.compile_graph_from_text(word_ids := list(words))
        for s in range(b.num_states):
            assert all(x.ilabel != 0 for x in b.arcs(s)), "the compiled graph must be epsilon-free"
        for T in (40, 61):
            for g_src, g_other in ((a, b), (b, a)):
                ok, ali = equal_align(ifst=g_src, length=T, rand_seed=seed + 3, num_retries=10)
                if not words:
                    continue
                assert ok
                ca, cb = _walk(g_src, ali), _walk(g_other, ali)
                assert cb is not None and abs(ca - cb) < 1e-4, (words, ca, cb)
        assert word_ids == list(words)
    with pytest.raises(Exception):
        gc.compile_graph_from_text([5])                      # a word the lexicon FST does not have


def test_word_id_equal_to_a_disambiguation_phone_id_is_still_a_word():
    """disambig_syms are phone-table ids (csrc/training-graph-compiler.cc:20-140 applies them to the INPUT side only); word ids come
    from another table and may carry the same numbers.  Word 7 below collides with the disambiguation phone #0 = 7: its arc must
    stay a word arc -- the transcript [1, 7] compiles, and the graph of [1, 2] must not accept a path through word 7."""
    from kaldi_hmm_gmm_amd import TrainingGraphCompiler, TrainingGraphCompilerOptions, equal_align, make_lexicon_fst_with_silence

    topo, cd, tm, _ = _setup(True)
    lex = {1: [(1.0, [2])], 2: [(1.0, [3])], 7: [(1.0, [4])]}
    L = make_lexicon_fst_with_silence(lex, sil_phone=1, sil_prob=0.5)
    gc = TrainingGraphCompiler(trans_model=tm, ctx_dep=cd, lex_fst=L, disambig_syms=[7, 8, 9], opts=TrainingGraphCompilerOptions())
    g17 = gc.compile_graph_from_text([1, 7])
    ok, ali17 = equal_align(ifst=g17, length=50, rand_seed=1, num_retries=10)
    assert ok and _walk(g17, ali17) is not None
    phones17 = {tm.transition_id_to_phone(t) for t in ali17}
    assert {2, 4} <= phones17 and 3 not in phones17
    g12 = gc.compile_graph_from_text([1, 2])
    assert _walk(g12, ali17) is None                         # word 7's phones are not a free detour of the graph for [1, 2]
    ok, ali12 = equal_align(ifst=g12, length=50, rand_seed=1, num_retries=10)
    assert ok and 4 not in {tm.transition_id_to_phone(t) for t in ali12}
    # every phone on any arc of the graph for [1, 2] belongs to words 1, 2 or to silence
    for s in range(g12.num_states):
        for a in g12.arcs(s):
            assert tm.transition_id_to_phone(a.ilabel) in (1, 2, 3)
