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
    """csrc/training-graph-compiler.h:32-40"""

    def __init__(self, transition_scale: float = 1.0, self_loop_scale: float = 1.0, reorder: bool = True):
        self.transition_scale = transition_scale
        self.self_loop_scale = self_loop_scale
        self.reorder = reorder


class TrainingGraphCompiler:
    """``lexicon``: word-id -> list of (probability, [phone ids]) pronunciations (scripts/prepare_lang.py
    Lexiconp).  ``sil_phone`` != None adds the optional silence of make_lexicon_fst_with_silence
    (before the first word and after every word, probability ``sil_prob``)."""

    def __init__(self, trans_model, ctx_dep, lexicon: Dict[int, List[Tuple[float, List[int]]]], sil_phone: Optional[int] = None,
                 sil_prob: float = 0.5, opts: Optional[TrainingGraphCompilerOptions] = None):
        if ctx_dep.context_width != 1 or ctx_dep.central_position != 0:
            raise KhgError("TrainingGraphCompiler: only monophone context (N=1, P=0) is supported")
        self.tm = trans_model
        self.ctx_dep = ctx_dep
        self.lexicon = lexicon
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

    def compile_graph_from_text(self, transcript: Sequence[int]) -> StdVectorFst:
        """csrc/training-graph-compiler.cc:65-141 for a linear word sequence."""
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


def equal_align(fst: StdVectorFst, length: int, rand_seed: int = 3, num_retries: int = 10):
    """Kaldi EqualAlign as used through kaldifst.equal_align (egs/yesno/train.py:86-108): a random
    path from the start to a final state, then the remaining frames spread as evenly as possible
    over the self-loops on that path.  Returns (ok, alignment) -- alignment = list of ``length``
    ilabels.  Only the contract is pinned by the reference (a valid path of exactly ``length``
    transition-ids, scripts/test_training_graph_compiler.py:85-105); the random choices come from a
    numpy Generator seeded with ``rand_seed`` instead of libc rand()."""
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
