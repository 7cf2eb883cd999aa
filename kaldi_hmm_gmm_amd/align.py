"""Alignment API -- mirrors of csrc/decoder-wrappers.{h,cc} (AlignConfig, AlignUtteranceWrapper),
csrc/faster-decoder.h (FasterDecoderOptions), csrc/decodable-am-diag-gmm.h and
csrc/hmm-utils.cc:465-493 (AddTransitionProbs), with their pybind names
(python/csrc/decoder-wrappers.cc, decodable-am-diag-gmm.cc, hmm-utils.cc).  The work is done by
K1 (log-likes) + K2 (Viterbi) through the C-ABI, for one utterance or a whole batch."""
from typing import List, Sequence

import numpy as np

from . import _gpu
from ._lib import KhgError
from .device import ALIGN_ERROR, ALIGN_RETRIED, INT32_MAX, DeviceModel, DeviceTransitions, UtteranceSet
from .diag_gmm import AmDiagGmm
from .fst import StdVectorFst, concat_graphs, modify_graph_for_careful_alignment
from .transition_model import TransitionModel


class AlignConfig:   # csrc/decoder-wrappers.h:23-37
    def __init__(self, beam: float = 200.0, retry_beam: float = 0.0, careful: bool = False):
        self.beam, self.retry_beam, self.careful = beam, retry_beam, careful


class FasterDecoderOptions:   # csrc/faster-decoder.h:24-63
    def __init__(self, beam: float = 16.0, max_active: int = INT32_MAX, min_active: int = 20,
                 beam_delta: float = 0.5, hash_ratio: float = 2.0):
        self.beam, self.max_active, self.min_active = beam, max_active, min_active
        self.beam_delta, self.hash_ratio = beam_delta, hash_ratio

    def __str__(self):
        return (f"FasterDecoderOptions(beam={self.beam:g}, max_active={self.max_active}, min_active={self.min_active}, "
                f"beam_delta={self.beam_delta:g}, hash_ratio={self.hash_ratio:g})")


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


class DecodableAmDiagGmmUnmapped:
    """csrc/decodable-am-diag-gmm.h:30-78: (frame, pdf-id + 1) -> log-likelihood.  Scores for every
    pdf are produced by one K1 launch on first use and kept (the reference's one-frame cache)."""

    def __init__(self, am: AmDiagGmm, feats, log_sum_exp_prune: float = -1.0):
        self._am = am
        self._feats = np.array(feats, np.float32, copy=True)
        if self._feats.ndim != 2:
            raise KhgError("feats must be a 2-D float matrix")
        self._ll = None

    def _scores(self):
        if self._ll is None:
            go, gc, _, miv, iv = self._am.flat()
            if self._am.dim != self._feats.shape[1]:
                raise KhgError(f"Dim mismatch: data dim = {self._feats.shape[1]} vs. model dim = {self._am.dim}")
            self._ll = _gpu.loglikes(go, gc, miv, iv, self._feats, np.arange(self._am.num_pdfs))
        return self._ll

    def log_likelihood(self, frame: int, index: int) -> float:
        return self._zero_based(frame, index - 1)

    def _zero_based(self, frame: int, state: int) -> float:
        if not 0 <= frame < self.num_frames_ready():
            raise KhgError("frame < NumFramesReady() assertion failed")
        if not 0 <= state < self._am.num_pdfs:
            raise KhgError("Likely graph/model mismatch, e.g. using wrong HCLG.fst")
        return float(self._scores()[state, frame])

    def num_frames_ready(self) -> int:
        return self._feats.shape[0]

    def num_indices(self) -> int:
        return self._am.num_pdfs

    def is_last_frame(self, frame: int) -> bool:
        if not frame < self.num_frames_ready():
            raise KhgError("frame < NumFramesReady() assertion failed")
        return frame == self.num_frames_ready() - 1


class DecodableAmDiagGmmScaled(DecodableAmDiagGmmUnmapped):
    """csrc/decodable-am-diag-gmm.h:83-103: scale * LL(frame, TransitionIdToPdf(tid))."""

    def __init__(self, am: AmDiagGmm, tm: TransitionModel, feats, scale: float, log_sum_exp_prune: float = -1.0):
        super().__init__(am, feats, log_sum_exp_prune)
        self._tm = tm
        self._scale = float(scale)

    @property
    def transition_model(self):
        return self._tm

    def log_likelihood(self, frame: int, tid: int) -> float:
        return float(np.float32(self._scale) * np.float32(self._zero_based(frame, self._tm.transition_id_to_pdf(tid))))

    def num_indices(self) -> int:
        return self._tm.num_transition_ids


def align_batch(am: AmDiagGmm, tm: TransitionModel, fsts: List[StdVectorFst], feats_list: List[np.ndarray],
                config: AlignConfig, acoustic_scale: float, trans_cost=None, decoder_opts: FasterDecoderOptions = None,
                return_scores: bool = False):
    """Batched AlignUtteranceWrapper: all utterances in one K1 + K2 pass.  `fsts` already carry their
    final arc weights unless `trans_cost` (per-tid additive cost, see TransitionModel.scaled_trans_cost)
    is given, in which case it is added on the device (the resident graphs stay unscaled)."""
    if (config.retry_beam != 0 and config.retry_beam <= config.beam) or config.beam <= 0.0:
        raise KhgError(f"Beams do not make sense: beam {config.beam}, retry-beam {config.retry_beam}")
    ctx = _gpu.default_context()
    if config.careful:
        fsts = [f.copy() for f in fsts]
        for f in fsts:
            if f.start != -1:
                modify_graph_for_careful_alignment(f)
    go, gc, _, miv, iv = am.flat()
    dm = DeviceModel(ctx, go, gc, miv, iv)
    dt = DeviceTransitions(ctx, np.asarray(tm.transition_id_to_pdf_array(), np.int32))
    dt.set_trans_cost(trans_cost)
    frame_off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats_list])]).astype(np.int64)
    feats = np.concatenate([np.asarray(f, np.float32).reshape(-1, am.dim) for f in feats_list]) if frame_off[-1] else \
        np.zeros((0, am.dim), np.float32)
    if feats.shape[0] == 0:
        feats = np.zeros((1, am.dim), np.float32)[:0]
    us = UtteranceSet(ctx, dt, frame_off, feats if feats.shape[0] else np.zeros((0, am.dim), np.float32), graphs=concat_graphs(fsts))
    us.loglikes(dm, reachable_only=True)      # only the cells a decoder token can read (khg_loglikes_reachable)
    o = decoder_opts or FasterDecoderOptions()
    res = us.align(dt, beam=config.beam, retry_beam=config.retry_beam, acoustic_scale=acoustic_scale,
                   careful=config.careful, max_active=o.max_active, min_active=o.min_active, beam_delta=o.beam_delta,
                   hash_ratio=o.hash_ratio)
    scores = us.download_loglikes() if return_scores else None
    poff, pdfl = us.pdf_lists() if return_scores else (None, None)
    out = []
    for u in range(len(fsts)):
        st = int(res["status"][u])
        ok = (st & ALIGN_ERROR) == 0
        out.append({
            "ok": ok, "retried": (st & ALIGN_RETRIED) != 0, "status": st,
            "alignment": res["ali"][frame_off[u]: frame_off[u + 1]].tolist() if ok else [],
            "words": res["words"][res["words_off"][u]: res["words_off"][u + 1]].tolist() if ok else [],
            "like": float(res["like"][u]) if ok else 0.0,
            "num_frames": int(frame_off[u + 1] - frame_off[u]),
        })
        if return_scores:
            out[-1]["loglikes"] = scores[u]                       # [npdf_u, T]
            out[-1]["pdfs"] = pdfl[poff[u]: poff[u + 1]]
    us.close(); dt.close(); dm.close()
    return out


def align_utterance_wrapper(config: AlignConfig, utt: str, acoustic_scale: float, fst: StdVectorFst,
                            decodable: DecodableAmDiagGmmScaled, num_done: int = 0, num_error: int = 0,
                            num_retried: int = 0, tot_like: float = 0.0, frame_count: int = 0):
    """python/csrc/decoder-wrappers.cc:25-47 -> (num_done, num_error, num_retried, tot_like, frame_count,
    alignment, words); counters are passed by value and returned incremented."""
    if not isinstance(decodable, DecodableAmDiagGmmScaled):
        raise KhgError("align_utterance_wrapper: the HIP path needs a DecodableAmDiagGmmScaled")
    if abs(decodable._scale - float(acoustic_scale)) > 0:
        # the reference scales scores by the decodable's scale and `like` by acoustic_scale; the scripts pass the same value
        raise KhgError("align_utterance_wrapper: decodable scale and acoustic_scale must agree on this path")
    if config.careful and fst.start != -1:
        modify_graph_for_careful_alignment(fst)      # the reference mutates the caller's fst (decoder-wrappers.cc:43-45)
        config = AlignConfig(config.beam, config.retry_beam, False)
    r = align_batch(decodable._am, decodable._tm, [fst], [decodable._feats], config, acoustic_scale)[0]
    if r["retried"]:
        num_retried += 1
    if not r["ok"]:
        return num_done, num_error + 1, num_retried, tot_like, frame_count, [], []
    return (num_done + 1, num_error, num_retried, tot_like + r["like"], frame_count + r["num_frames"], r["alignment"],
            r["words"])


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
