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


def add_transition_probs(trans_model: TransitionModel, transition_scale: float, self_loop_scale: float,
                         fst: StdVectorFst, disambig_syms: Sequence[int] = ()) -> None:
    """csrc/hmm-utils.cc:465-493: arc.weight (x)= -scaled transition log-prob, in place."""
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
                config: AlignConfig, acoustic_scale: float, trans_cost=None, decoder_opts: FasterDecoderOptions = None):
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
    us.loglikes(dm)
    o = decoder_opts or FasterDecoderOptions()
    res = us.align(dt, beam=config.beam, retry_beam=config.retry_beam, acoustic_scale=acoustic_scale,
                   careful=config.careful, max_active=o.max_active, min_active=o.min_active, beam_delta=o.beam_delta,
                   hash_ratio=o.hash_ratio)
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
