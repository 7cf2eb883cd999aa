"""Alignment API -- AlignConfig, FasterDecoderOptions, DecodableAmDiagGmmUnmapped / Scaled, align_batch and
align_utterance_wrapper are the C++ of csrc/khg_host_align.{hpp,cpp} (mirrors of csrc/decoder-wrappers.{h,cc},
csrc/faster-decoder.h, csrc/decodable-am-diag-gmm.h) under their pybind names (python/csrc/decoder-wrappers.cc,
decodable-am-diag-gmm.cc, faster-decoder.cc); the work is done by K1 (log-likes) + K2 (Viterbi) through the C-ABI, for one
utterance or a whole batch.  Python keeps what needs the graph container (the reference's kaldifst VectorFst is the Python
StdVectorFst here): AddTransitionProbs (csrc/hmm-utils.cc:465-493) and FasterDecoder.get_best_path's linear lattice."""
from typing import List, Sequence

import numpy as np

from . import device  # noqa: F401
from ._kaldi_hmm_gmm_amd import (AlignConfig, DecodableAmDiagGmmScaled, DecodableAmDiagGmmUnmapped, DecodableInterface,  # noqa: F401
                                 FasterDecoderOptions, align_batch, align_utterance_wrapper)
from ._lib import KhgError
from .device import ALIGN_ERROR, ALIGN_RETRIED, INT32_MAX  # noqa: F401
from .fst import StdVectorFst
from .transition_model import TransitionModel


def add_transition_probs(trans_model: TransitionModel, disambig_syms: Sequence[int] = (), transition_scale: float = None,
                         self_loop_scale: float = None, fst: StdVectorFst = None) -> None:
    """csrc/hmm-utils.cc:465-493: arc.weight (x)= -scaled transition log-prob, in place.  Argument order and
    names of python/csrc/hmm-utils.cc:14-19 (disambig_syms defaults to empty; the rest are required)."""
    if transition_scale is None or self_loop_scale is None or fst is None:
        raise TypeError("add_transition_probs(): transition_scale, self_loop_scale and fst are required")
    dis = sorted(disambig_syms)
    if list(dis) != list(disambig_syms):
        raise KhgError("IsSortedAndUniq(disambig_syms) assertion failed")
    cost = trans_model.scaled_trans_cost(transition_scale, self_loop_scale)
    nt = trans_model.num_transition_ids
    for s in range(fst.num_states):
        for a in fst.arcs(s):
            if 1 <= a.ilabel <= nt:
                a.weight = float(np.float32(np.float32(a.weight) + cost[a.ilabel]))
            elif a.ilabel != 0 and a.ilabel not in dis:
                raise KhgError(f"AddTransitionProbs: invalid symbol {a.ilabel} on graph input side.")


class LatticeWeight:
    """kaldifst LatticeWeight (graph cost, acoustic cost); Times adds component-wise."""

    def __init__(self, value1: float = 0.0, value2: float = 0.0):
        self.value1, self.value2 = float(value1), float(value2)

    def __repr__(self):
        return f"LatticeWeight({self.value1}, {self.value2})"


class LatticeArc:
    def __init__(self, ilabel: int, olabel: int, weight: LatticeWeight, nextstate: int):
        self.ilabel, self.olabel, self.weight, self.nextstate = int(ilabel), int(olabel), weight, int(nextstate)


class LinearLattice:
    """The linear fst::VectorFst<LatticeArc> FasterDecoder::GetBestPath returns: state i has the single arc
    arcs[i] to state i+1; the last state is final with `final`."""

    def __init__(self):
        self.arcs: List[LatticeArc] = []
        self.final = LatticeWeight()
        self.start = -1

    @property
    def num_states(self) -> int:
        return 0 if self.start < 0 else len(self.arcs) + 1

    def get_linear_symbol_sequence(self):
        """kaldifst GetLinearSymbolSequence -> (ok, ilabels != 0, olabels != 0, total LatticeWeight)."""
        if self.start < 0:
            return False, [], [], LatticeWeight()
        w = LatticeWeight(self.final.value1, self.final.value2)
        for a in self.arcs:
            w = LatticeWeight(w.value1 + a.weight.value1, w.value2 + a.weight.value2)
        return True, [a.ilabel for a in self.arcs if a.ilabel], [a.olabel for a in self.arcs if a.olabel], w


class FasterDecoder:
    """python/csrc/faster-decoder.cc:33-53 on the GPU path: ``decode`` runs K1 + K2 for the utterance of a
    DecodableAmDiagGmmScaled with the FasterDecoderOptions' beam / max_active / min_active / beam_delta /
    hash_ratio (no retry), ``get_best_path`` rebuilds the linear lattice of csrc/faster-decoder.cc:355-423
    from the alignment: arc weights (graph cost, acoustic cost) per token, final weight, true epsilons
    removed.  Whole utterances only: ``advanced_decoding`` with a frame limit is not supported."""

    def __init__(self, fst: StdVectorFst, config: FasterDecoderOptions):
        self._fst = fst
        self.set_options(config)
        self._res = None
        self._nframes = -1

    def set_options(self, config: FasterDecoderOptions):
        if not (config.hash_ratio >= 1.0) or not (config.max_active > 1) or not (0 <= config.min_active < config.max_active):
            raise KhgError("FasterDecoder: bad options (hash_ratio >= 1, max_active > 1, 0 <= min_active < max_active)")
        self._cfg = config

    def init_decoding(self):
        if self._fst.start < 0:
            raise KhgError("start_state != fst::kNoStateId assertion failed")
        self._res = None
        self._nframes = 0

    def decode(self, decodable: "DecodableAmDiagGmmScaled"):
        self.init_decoding()
        self.advanced_decoding(decodable)

    def advanced_decoding(self, decodable: "DecodableAmDiagGmmScaled", max_num_frames: int = -1):
        if not isinstance(decodable, DecodableAmDiagGmmScaled):
            raise KhgError("FasterDecoder: the HIP path needs a DecodableAmDiagGmmScaled")
        if max_num_frames >= 0 and max_num_frames < decodable.num_frames_ready():
            raise KhgError("FasterDecoder.advanced_decoding: partial decoding (max_num_frames) is not supported on the HIP path")
        if self._nframes < 0:
            raise KhgError("num_frames_decoded_ >= 0 assertion failed: call init_decoding() first")
        cfg = AlignConfig(beam=self._cfg.beam, retry_beam=0.0)
        self._dec = decodable
        self._res = align_batch(decodable._am, decodable._tm, [self._fst], [decodable._feats], cfg, decodable._scale,
                                decoder_opts=self._cfg, return_scores=True)[0]
        self._nframes = decodable.num_frames_ready()

    def num_frames_decoded(self) -> int:
        return self._nframes

    def reached_final(self) -> bool:
        return bool(self._res and self._res["ok"])

    def get_best_path(self, use_final_probs: bool = True):
        lat = LinearLattice()
        if not self.reached_final():
            # the reference would fall back to the best non-final token; the HIP kernels keep no such token
            return False, lat
        r = self._res
        ali = r["alignment"]
        tm = self._dec._tm
        col = {int(p): i for i, p in enumerate(r["pdfs"])}
        scale = np.float32(self._dec._scale)
        ac = [float(-np.float32(scale * np.float32(r["loglikes"][col[tm.transition_id_to_pdf(t)], i]))) for i, t in enumerate(ali)]
        # cheapest path through the graph with exactly this input-label sequence (= the decoder's best path)
        fst = self._fst
        INF = float("inf")
        layer = {fst.start: (0.0, None)}
        back = []          # per step: dict state -> (prev_state, arc, is_eps)

        def closure(layer, bp):
            stack = list(layer)
            while stack:
                s = stack.pop()
                c = layer[s][0]
                for a in fst.arcs(s):
                    if a.ilabel == 0:
                        v = c + a.weight
                        if v < layer.get(a.nextstate, (INF,))[0]:
                            layer[a.nextstate] = (v, None)
                            bp[a.nextstate] = (s, a)
                            stack.append(a.nextstate)

        eps_bp = {}
        closure(layer, eps_bp)
        back.append(({}, eps_bp))
        for i, t in enumerate(ali):
            nxt, bp = {}, {}
            for s, (c, _) in layer.items():
                for a in fst.arcs(s):
                    if a.ilabel == t:
                        v = c + a.weight + ac[i]
                        if v < nxt.get(a.nextstate, (INF,))[0]:
                            nxt[a.nextstate] = (v, None)
                            bp[a.nextstate] = (s, a)
            eps_bp = {}
            closure(nxt, eps_bp)
            back.append((bp, eps_bp))
            layer = nxt
        best, bs = INF, None
        for s, (c, _) in layer.items():
            if fst.is_final(s) and c + fst.final(s) < best:
                best, bs = c + fst.final(s), s
        if bs is None:
            return False, lat
        arcs = []
        s = bs
        for i in range(len(ali), -1, -1):
            bp, eps_bp = back[i]
            while s in eps_bp:                       # epsilon hops inside layer i
                ps, a = eps_bp[s]
                arcs.append((a, None))
                s = ps
            if i > 0:
                ps, a = bp[s]
                arcs.append((a, ac[i - 1]))
                s = ps
        arcs.reverse()
        lat.start = 0
        carry = LatticeWeight()
        for a, acost in arcs:
            w = LatticeWeight(a.weight + carry.value1, (acost or 0.0) + carry.value2)
            if a.ilabel == 0 and a.olabel == 0:      # RemoveEpsLocal on a linear lattice: fold true epsilons forward
                carry = w
                continue
            carry = LatticeWeight()
            lat.arcs.append(LatticeArc(a.ilabel, a.olabel, w, len(lat.arcs) + 1))
        fw = fst.final(bs) if use_final_probs else 0.0
        lat.final = LatticeWeight(fw + carry.value1, carry.value2)
        return True, lat
