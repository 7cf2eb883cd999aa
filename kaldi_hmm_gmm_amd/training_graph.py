"""Training-graph builder and equal-align for linear transcripts.

Replaces, for monophone trees and word-level transcripts, what the reference gets from kaldifst:

  * ``generate_hmm_topo``                     scripts/prepare_lang.py:514-600
  * ``Lexicon`` + optional silence            scripts/prepare_lang.py:329-456 (make_lexicon_fst_with_silence)
  * ``TrainingGraphCompiler``                 csrc/training-graph-compiler.cc:65-141 (CompileGraphFromText)
        H expansion                           csrc/hmm-utils.cc:40-158  (GetHmmAsFsa: no self-loops, arc cost
                                              -transition_scale * log(p / (1 - p_selfloop)))
        self-loops, "reorder"                 csrc/hmm-utils.cc:293-369 (AddSelfLoopsReorder)
  * ``equal_align``                           kaldifst.equal_align as called by egs/yesno/train.py:86-108

The reference builds H o C o L o G with generic FST algorithms (TableCompose, DeterminizeStarInLog,
MinimizeEncoded).  Here the same graph is built directly: the set of accepted transition-id sequences
and the total cost of every path are those of the reference's graph; state numbering and the placement
of weights along a path (log-semiring determinization may push them) are not, which only matters to
beam pruning.  The result is epsilon-free, so the Viterbi kernel's register-resident path applies.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import KhgError
from .fst import StdArc, StdVectorFst
from .hmm_topology import HmmTopology, kNoPdf


def generate_hmm_topo(non_sil_phones: List[int], sil_phone: int, num_non_sil_states: int = 3,
                      num_sil_states: int = 5) -> HmmTopology:
    """scripts/prepare_lang.py:514-600: left-to-right non-silence phones (self-loop 0.75 / forward
    0.25); silence: first state fans out to all but the last emitting state, middle states fully
    connected, last emitting state non-skippable."""
    s = "<Topology> <TopologyEntry> <ForPhones> " + " ".join(map(str, non_sil_phones)) + "\n</ForPhones> "
    for i in range(num_non_sil_states):
        s += f"<State> {i} <PdfClass> {i} <Transition> {i} 0.75 <Transition> {i + 1} 0.25 </State> "
    s += f"<State> {num_non_sil_states} </State> </TopologyEntry> "
    if num_sil_states > 1:
        transp = 1.0 / (num_sil_states - 1)
        s += f"<TopologyEntry> <ForPhones> {sil_phone} </ForPhones> <State> 0 <PdfClass> 0 "
        for i in range(num_sil_states - 1):
            s += f"<Transition> {i} {transp} "
        s += "</State> "
        for i in range(1, num_sil_states - 1):
            s += f"<State> {i} <PdfClass> {i} "
            for k in range(1, num_sil_states):
                s += f"<Transition> {k} {transp} "
            s += "</State> "
        s += (f"<State> {num_sil_states - 1} <PdfClass> {num_sil_states - 1} <Transition> {num_sil_states - 1} 0.75 "
              f"<Transition> {num_sil_states} 0.25 </State> <State> {num_sil_states} </State> </TopologyEntry> ")
    else:
        s += (f"<TopologyEntry> <ForPhones> {sil_phone} </ForPhones> <State> 0 <PdfClass> 0 <Transition> 0 0.75 "
              f"<Transition> 1 0.25 </State> <State> 1 </State> </TopologyEntry> ")
    s += "</Topology>"
    topo = HmmTopology()
    topo.read(s)
    return topo


class TrainingGraphCompilerOptions:
    """csrc/training-graph-compiler.h:32-40; python/csrc/training-graph-compiler.cc:17-27 (rm_eps is an attribute, not a
    constructor argument, there)."""

    def __init__(self, transition_scale: float = 1.0, self_loop_scale: float = 1.0, reorder: bool = True):
        self.transition_scale = transition_scale
        self.self_loop_scale = self_loop_scale
        self.rm_eps = False        # the graphs built here are epsilon-free either way
        self.reorder = reorder

    def __str__(self):
        return (f"TrainingGraphCompilerOptions(transition_scale={self.transition_scale}, self_loop_scale={self.self_loop_scale}, "
                f"rm_eps={self.rm_eps}, reorder={self.reorder})")


class TrainingGraphCompiler:
    """python/csrc/training-graph-compiler.cc:32-58: ``TrainingGraphCompiler(trans_model, ctx_dep, lex_fst, disambig_syms, opts)``.

    ``lex_fst`` is the lexicon, in one of two forms:
      * an ``StdVectorFst`` L (input labels = phones, output labels = words; scripts/prepare_lang.py:329-456 writes it) -- input
        labels listed in ``disambig_syms`` are treated as epsilons, like the reference's graph compilation removes them;
      * a dict word-id -> list of (probability, [phone ids]) pronunciations (scripts/prepare_lang.py Lexiconp); then
        ``sil_phone`` != None adds the optional silence of make_lexicon_fst_with_silence (before the first word and after
        every word, probability ``sil_prob``).
    """

    def __init__(self, trans_model, ctx_dep, lex_fst, disambig_syms: Optional[Sequence[int]] = None,
                 opts: Optional[TrainingGraphCompilerOptions] = None, *, sil_phone: Optional[int] = None, sil_prob: float = 0.5):
        if ctx_dep.context_width != 1 or ctx_dep.central_position != 0:
            raise KhgError("TrainingGraphCompiler: only monophone context (N=1, P=0) is supported")
        self.tm = trans_model
        self.ctx_dep = ctx_dep
        if isinstance(lex_fst, dict):
            self.lexicon, self.lex_fst = lex_fst, None
        else:
            self.lexicon, self.lex_fst = None, lex_fst.copy()       # the reference's binding copies too (lex_fst->Copy())
            if sil_phone is not None:
                raise KhgError("TrainingGraphCompiler: sil_phone belongs to the dict form; an L.fst carries its own silence arcs")
        self.disambig_syms = sorted(int(x) for x in (disambig_syms or []))
        self.sil_phone = sil_phone
        self.sil_prob = float(sil_prob)
        self.opts = opts or TrainingGraphCompilerOptions()
        if not self.opts.reorder:
            raise KhgError("TrainingGraphCompiler: reorder=False is not supported (the reference default is True)")
        self._hmm_cache = {}

    # ---- one phone as an FSA without self-loops (csrc/hmm-utils.cc:40-158) ----
    def _phone_hmm(self, phone: int):
        """-> (num_hmm_states, [(src_hmm_state, dst_hmm_state, tid, cost)]) ; last state is final."""
        if phone in self._hmm_cache:
            return self._hmm_cache[phone]
        topo = self.tm.topo
        entry = topo.topology_for_phone(phone)
        pdfs = []
        for pc in range(topo.num_pdf_classes(phone)):
            ok, pdf = self.ctx_dep.compute([phone], pc)
            if not ok:
                raise KhgError(f"GetHmmAsFsa: context-dependency object could not produce an answer: pdf-class = {pc} ctx-window = {phone}")
            pdfs.append(pdf)
        arcs = []
        for h, st in enumerate(entry):
            if st.forward_pdf_class == kNoPdf:
                if st.transitions:
                    raise KhgError("GetHmmAsFsa: non-emitting states with transitions are not supported")
                continue
            fpdf, spdf = pdfs[st.forward_pdf_class], pdfs[st.self_loop_pdf_class]
            ts = self.tm.tuple_to_transition_state(phone, h, fpdf, spdf)
            for idx, (dst, _) in enumerate(st.transitions):
                if dst == h:
                    continue  # self-loops are added later
                tid = self.tm.pair_to_transition_id(ts, idx)
                logp = self.tm.get_transition_log_prob_ignoring_self_loops(tid)
                # ApplyProbabilityScale(transition_scale) on the float32 arc weight
                arcs.append((h, dst, tid, float(np.float32(np.float32(-logp) * np.float32(self.opts.transition_scale)))))
        self._hmm_cache[phone] = (len(entry), arcs)
        return self._hmm_cache[phone]

    # ---- phone-level graph of L o G for a linear transcript (eps-free) ----
    def _phone_graph(self, transcript: Sequence[int]):
        """States: 0 = start, then per boundary k = 0..n a "loop" node (after k words) and, with
        silence, a "sil" node.  Returns (num_nodes, arcs[(src, dst, phone, olabel, cost)], final_node).
        Epsilon arcs of the lexicon FST at the start are folded into their successors."""
        n = len(transcript)
        sil = self.sil_phone is not None
        sil_cost = -math.log(self.sil_prob) if sil else 0.0
        no_sil_cost = -math.log(1.0 - self.sil_prob) if sil else 0.0
        nodes = 0

        def new():
            nonlocal nodes
            nodes += 1
            return nodes - 1

        start = new()
        loop = [new() for _ in range(n + 1)]
        silst = [new() for _ in range(n + 1)] if sil else []
        arcs = []
        word_arcs = [[] for _ in range(n)]          # arcs leaving loop[k] for word k (to replicate from start)
        for k, w in enumerate(transcript):
            if w not in self.lexicon:
                raise KhgError(f"TrainingGraphCompiler: word {w} is not in the lexicon")
            for prob, phones in self.lexicon[w]:
                if not phones:
                    raise KhgError("TrainingGraphCompiler: empty pronunciations are not supported")
                pron_cost = -math.log(float(prob))
                cur = loop[k]
                for i, ph in enumerate(phones):
                    first, last = i == 0, i == len(phones) - 1
                    base = pron_cost if first else 0.0
                    ol = w if first else 0
                    if not last:
                        nxt = new()
                        a = (cur, nxt, ph, ol, base)
                        arcs.append(a)
                        if first:
                            word_arcs[k].append(a)
                        cur = nxt
                    else:
                        a = (cur, loop[k + 1], ph, ol, base + no_sil_cost)
                        arcs.append(a)
                        if first:
                            word_arcs[k].append(a)
                        if sil:
                            a = (cur, silst[k + 1], ph, ol, base + sil_cost)
                            arcs.append(a)
                            if first:
                                word_arcs[k].append(a)
        if sil:
            for k in range(n + 1):
                arcs.append((silst[k], loop[k], self.sil_phone, 0, 0.0))
        # start: eps(no_sil_cost) -> loop[0]  and  eps(sil_cost) -> sil[0]   (prepare_lang.py:352-371), folded
        if sil:
            arcs.append((start, loop[0], self.sil_phone, 0, sil_cost))
        if n > 0:
            for (_, dst, ph, ol, c) in word_arcs[0]:
                arcs.append((start, dst, ph, ol, c + no_sil_cost))
        start_final = no_sil_cost if n == 0 else None
        return nodes, arcs, loop[n], start, start_final

    # ---- the same from an L.fst: L o (linear acceptor of the transcript), input epsilons / disambiguation symbols removed ----
    def _phone_graph_from_lfst(self, transcript: Sequence[int]):
        """-> (num_nodes, arcs[(src, dst, phone, olabel, cost)], finals{node: cost}, start).  Product of the lexicon FST with the
        word positions 0..n: an arc with output label 0 moves in L only, an arc whose output label is the next word moves in
        both (csrc/training-graph-compiler.cc:74-96 composes L with the transcript's linear acceptor); then every arc whose
        input label is 0 or a disambiguation symbol is folded into its successors (tropical epsilon removal)."""
        L, n = self.lex_fst, len(transcript)
        if L.start < 0:
            raise KhgError("TrainingGraphCompiler: empty lexicon FST")
        dis = set(self.disambig_syms)
        index, order = {}, []

        def get(ls, k):
            if (ls, k) not in index:
                index[(ls, k)] = len(order)
                order.append((ls, k))
            return index[(ls, k)]

        get(L.start, 0)
        raw = []                       # (src, dst, phone (0 = epsilon), olabel, cost)
        finals = {}
        i = 0
        while i < len(order):
            ls, k = order[i]
            src = i
            i += 1
            if k == n and L.is_final(ls):
                finals[src] = float(L.final(ls))
            for a in L.arcs(ls):
                # disambig_syms are PHONE-table ids: they name input labels only (csrc/training-graph-compiler.cc:20-140 removes
                # them from the phone side after composing L with the transcript on the real word ids) -- an output label is a word
                # id from another table and is epsilon only when it is 0, whatever number a disambiguation phone happens to carry
                ph = 0 if (a.ilabel == 0 or a.ilabel in dis) else a.ilabel
                if a.olabel == 0:
                    raw.append((src, get(a.nextstate, k), ph, 0, float(a.weight)))
                elif k < n and a.olabel == transcript[k]:
                    raw.append((src, get(a.nextstate, k + 1), ph, a.olabel, float(a.weight)))
        nodes = len(order)
        # epsilon removal: eps-closure (cheapest cost, at most one word label) of every node, then re-emit the closure's real arcs
        out = [[] for _ in range(nodes)]
        for r in raw:
            out[r[0]].append(r)
        arcs, new_finals = [], {}
        for s0 in range(nodes):
            best = {s0: (0.0, 0)}      # node -> (cost, carried olabel)
            stack = [s0]
            while stack:
                s = stack.pop()
                c, ol = best[s]
                for (_, d, ph, ol2, w) in out[s]:
                    if ph != 0:
                        continue
                    if ol and ol2:
                        raise KhgError("TrainingGraphCompiler: two word labels on one chain of epsilon arcs of the lexicon FST")
                    v = (c + w, ol or ol2)
                    if d not in best or v[0] < best[d][0]:
                        best[d] = v
                        stack.append(d)
            for s, (c, ol) in best.items():
                if s in finals:
                    if ol:
                        raise KhgError("TrainingGraphCompiler: a word label on an epsilon arc into a final state of the lexicon FST")
                    v = c + finals[s]
                    if s0 not in new_finals or v < new_finals[s0]:
                        new_finals[s0] = v
                for (_, d, ph, ol2, w) in out[s]:
                    if ph == 0:
                        continue
                    if ol and ol2:
                        raise KhgError("TrainingGraphCompiler: a word label on an epsilon arc before a labelled phone arc")
                    arcs.append((s0, d, ph, ol or ol2, c + w))
        # keep what the start can reach (nodes that were only reachable through epsilons are now dead)
        reach, stack = {0}, [0]
        by_src = {}
        for a in arcs:
            by_src.setdefault(a[0], []).append(a)
        while stack:
            s = stack.pop()
            for a in by_src.get(s, []):
                if a[1] not in reach:
                    reach.add(a[1])
                    stack.append(a[1])
        arcs = [a for a in arcs if a[0] in reach]
        new_finals = {s: c for s, c in new_finals.items() if s in reach}
        return nodes, arcs, new_finals, 0

    def compile_graph_from_text(self, transcript: Sequence[int]) -> StdVectorFst:
        """csrc/training-graph-compiler.cc:65-141 for a linear word sequence."""
        if self.lex_fst is not None:
            nodes, parcs, finals, start = self._phone_graph_from_lfst([int(w) for w in transcript])
            if not finals:
                raise KhgError("succeeded assertion failed: the lexicon FST does not accept the transcript")   # KHG_ASSERT(succeeded)
            return self._expand(nodes, parcs, finals, start)
        nodes, parcs, final_node, start, start_final = self._phone_graph(list(transcript))
        finals = {final_node: 0.0}
        if start_final is not None:
            finals[start] = start_final
        return self._expand(nodes, parcs, finals, start)

    def compile_word_loop_graph(self, word_probs: Optional[Dict[int, float]] = None) -> StdVectorFst:
        """A decoding graph for the same lexicon: a unigram word loop (every word of the lexicon, probability
        ``word_probs[w]`` or uniform) with the lexicon's optional silence between words -- H o C o L o G for
        G = a one-state unigram, built with the same expansion as the training graphs (transition-ids on the
        input side, word-ids on the output side, "reorder" self-loops, epsilon-free).  What egs/yesno/decode.py
        decodes with (there: HLG from the lang directory)."""
        if self.lexicon is None:
            raise KhgError("compile_word_loop_graph: needs the dict form of the lexicon")
        words = sorted(self.lexicon)
        if not words:
            raise KhgError("compile_word_loop_graph: empty lexicon")
        wp = word_probs or {w: 1.0 / len(words) for w in words}
        sil = self.sil_phone is not None
        sil_cost = -math.log(self.sil_prob) if sil else 0.0
        no_sil_cost = -math.log(1.0 - self.sil_prob) if sil else 0.0
        # nodes: start (before the leading-silence decision), loop (between words), sil (a word was followed by silence)
        nodes = 3 if sil else 1
        start, loop, silst = (0, 1, 2) if sil else (0, 0, 0)

        def new():
            nonlocal nodes
            nodes += 1
            return nodes - 1

        arcs = []
        for src, extra in (((loop, 0.0), (start, no_sil_cost)) if sil else ((loop, 0.0),)):
            for w in words:
                for prob, phones in self.lexicon[w]:
                    if not phones:
                        raise KhgError("TrainingGraphCompiler: empty pronunciations are not supported")
                    base0 = -math.log(float(prob)) - math.log(float(wp[w])) + extra
                    cur = src
                    for i, ph in enumerate(phones):
                        first, last = i == 0, i == len(phones) - 1
                        base, ol = (base0 if first else 0.0), (w if first else 0)
                        if not last:
                            nxt = new()
                            arcs.append((cur, nxt, ph, ol, base))
                            cur = nxt
                        else:
                            arcs.append((cur, loop, ph, ol, base + no_sil_cost))
                            if sil:
                                arcs.append((cur, silst, ph, ol, base + sil_cost))
        finals = {loop: 0.0}
        if sil:
            arcs.append((silst, loop, self.sil_phone, 0, 0.0))
            arcs.append((start, loop, self.sil_phone, 0, sil_cost))   # the lexicon's start epsilons (prepare_lang.py:352-371), folded
            finals[start] = no_sil_cost
        return self._expand(nodes, arcs, finals, start)

    def _expand(self, nodes: int, parcs, finals: Dict[int, float], start: int) -> StdVectorFst:
        """Phone-level graph -> transition-id graph: GetHmmAsFsa per phone arc, then
        MakePrecedingInputSymbolsSameClass + AddSelfLoopsReorder (csrc/hmm-utils.cc:293-369)."""
        # --- expand phones into transition-id arcs (no self-loops yet) ---
        out = [[] for _ in range(nodes)]      # per state: (dst, tid, olabel, cost)

        def new_state():
            out.append([])
            return len(out) - 1

        for (src, dst, ph, ol, cost) in parcs:
            nst, harcs = self._phone_hmm(ph)
            ids = {0: src, nst - 1: dst}
            for (h, d, tid, c) in harcs:
                for x in (h, d):
                    if x not in ids:
                        ids[x] = new_state()
                first = h == 0
                out[ids[h]].append((ids[d], tid, ol if first else 0, float(np.float32(c + (cost if first else 0.0)))))
        # a state copy per class (transition-state) of its incoming arcs
        fst = StdVectorFst()
        index = {}
        order = []

        def get(state, cls):
            key = (state, cls)
            if key not in index:
                index[key] = fst.add_state()
                order.append(key)
            return index[key]

        get(start, 0)
        fst.start = 0
        sl = np.float32(self.opts.self_loop_scale)
        i = 0
        while i < len(order):
            state, cls = order[i]
            sid = index[(state, cls)]
            i += 1
            mult = 0.0
            if cls > 0:
                mult = float(np.float32(-np.float32(self.tm.get_non_self_loop_log_prob(cls)) * sl))
            if state in finals:
                fst.set_final(sid, float(np.float32(np.float32(finals[state]) + np.float32(mult))))
            for (dst, tid, ol, c) in out[state]:
                ts = self.tm.transition_id_to_transition_state(tid)
                fst.add_arc(sid, StdArc(tid, ol, float(np.float32(np.float32(c) + np.float32(mult))), get(dst, ts)))
            if cls > 0:
                loop_tid = self.tm.self_loop_of(cls)
                if loop_tid != 0:
                    lp = np.float32(self.tm.get_transition_log_prob(loop_tid))
                    fst.add_arc(sid, StdArc(loop_tid, 0, float(np.float32(-lp * sl)), sid))
        return fst

    def compile_graphs_from_text(self, transcripts: Sequence[Sequence[int]]) -> List[StdVectorFst]:
        return [self.compile_graph_from_text(t) for t in transcripts]


def make_lexicon_fst_with_silence(lexicon: Dict[int, List[Tuple[float, List[int]]]], sil_phone: int, sil_prob: float = 0.5,
                                  sil_disambig: Optional[int] = None) -> StdVectorFst:
    """The lexicon FST L of scripts/prepare_lang.py:329-456 over integer ids: ``lexicon`` maps word-id -> [(probability, [phone
    ids, possibly ending in a disambiguation symbol id])].  State 0 = start, 1 = loop (final), 2 = silence; the optional silence
    (probability ``sil_prob``) sits before the first word and after every word.  What egs/yesno/train.py:58-63 hands to
    TrainingGraphCompiler as ``lex_fst``."""
    sil_cost = -math.log(sil_prob)
    no_sil_cost = -math.log(1.0 - sil_prob)
    fst = StdVectorFst()
    start_state, loop_state, sil_state = fst.add_state(), fst.add_state(), fst.add_state()
    fst.start = start_state
    fst.set_final(state=loop_state, weight=0)
    fst.add_arc(state=start_state, arc=StdArc(ilabel=0, olabel=0, weight=no_sil_cost, nextstate=loop_state))
    fst.add_arc(state=start_state, arc=StdArc(ilabel=0, olabel=0, weight=sil_cost, nextstate=sil_state))
    if sil_disambig is None:
        fst.add_arc(state=sil_state, arc=StdArc(ilabel=sil_phone, olabel=0, weight=0, nextstate=loop_state))
    else:
        d = fst.add_state()
        fst.add_arc(state=sil_state, arc=StdArc(ilabel=sil_phone, olabel=0, weight=0, nextstate=d))
        fst.add_arc(state=d, arc=StdArc(ilabel=sil_disambig, olabel=0, weight=0, nextstate=loop_state))
    for word in lexicon:
        for prob, phones in lexicon[word]:
            pron_cost = -math.log(float(prob))
            cur = loop_state
            for i in range(len(phones) - 1):
                nxt = fst.add_state()
                fst.add_arc(state=cur, arc=StdArc(ilabel=phones[i], olabel=word if i == 0 else 0, weight=pron_cost if i == 0 else 0, nextstate=nxt))
                cur = nxt
            i = len(phones) - 1                    # -1 for an empty pronunciation
            il = phones[i] if i >= 0 else 0
            ol = word if i <= 0 else 0
            w = pron_cost if i <= 0 else 0
            fst.add_arc(state=cur, arc=StdArc(ilabel=il, olabel=ol, weight=no_sil_cost + w, nextstate=loop_state))
            fst.add_arc(state=cur, arc=StdArc(ilabel=il, olabel=ol, weight=sil_cost + w, nextstate=sil_state))
    return fst


def equal_align(ifst: StdVectorFst, length: int, rand_seed: int = 3, num_retries: int = 10):
    """Kaldi EqualAlign as used through kaldifst.equal_align (egs/yesno/train.py:86-108): a random
    path from the start to a final state, then the remaining frames spread as evenly as possible
    over the self-loops on that path.  Returns (ok, alignment) -- alignment = list of ``length``
    ilabels.  Only the contract is pinned by the reference (a valid path of exactly ``length``
    transition-ids, scripts/test_training_graph_compiler.py:85-105); the random choices come from a
    numpy Generator seeded with ``rand_seed`` instead of libc rand()."""
    fst = ifst        # (kaldifst.equal_align's keyword, egs/yesno/train.py:88-93)
    if fst.start < 0:
        return False, []
    rng = np.random.default_rng(rand_seed)
    for _ in range(max(1, num_retries)):
        path = []           # (state, arc or None for "stay")
        s = fst.start
        nlabels = 0
        ok = True
        for _step in range(max(length * 4 + 16, 64)):
            arcs = [a for a in fst.arcs(s) if a.nextstate != s]
            nopt = len(arcs) + (1 if fst.is_final(s) else 0)
            if nopt == 0:
                ok = False
                break
            k = int(rng.integers(nopt))
            if k == len(arcs):
                break                      # stop at this final state
            a = arcs[k]
            path.append((s, a))
            if a.ilabel != 0:
                nlabels += 1
            s = a.nextstate
        else:
            ok = False
        if not ok or not fst.is_final(s) or nlabels > length:
            continue
        # states on the path that own a self-loop with an input label, in path order
        loops = []
        for pos, (_, a) in enumerate(path):
            for sa in fst.arcs(a.nextstate):
                if sa.nextstate == a.nextstate and sa.ilabel != 0:
                    loops.append((pos, sa.ilabel))
                    break
        extra = length - nlabels
        if extra > 0 and not loops:
            continue
        counts = [0] * len(loops)
        if loops:
            base, rem = divmod(extra, len(loops))
            counts = [base + (1 if j < rem else 0) for j in range(len(loops))]
        at = {pos: (lab, counts[j]) for j, (pos, lab) in enumerate(loops)}
        ali = []
        for pos, (_, a) in enumerate(path):
            if a.ilabel != 0:
                ali.append(a.ilabel)
            if pos in at:
                ali.extend([at[pos][0]] * at[pos][1])
        if len(ali) == length:
            return True, ali
    return False, []
