"""TransitionModel -- host-side mirror of csrc/transition-model.{h,cc} / transition-information.h
(python/csrc/transition-model.cc, transition-information.cc).  Integer tables are built here
(init-time glue); the M-step update runs in C++ (khg_transition_mle_update)."""
import ctypes as C
from typing import List

import numpy as np

from . import _lib
from ._lib import KhgError, check, lib, ptr
from .hmm_topology import HmmTopology, kNoPdf

_libm = C.CDLL("libm.so.6")
_libm.expf.restype = C.c_float
_libm.expf.argtypes = [C.c_float]
_libm.logf.restype = C.c_float
_libm.logf.argtypes = [C.c_float]


def _libm_expf(x: float) -> float:
    return _libm.expf(x)


def _f32_sum(v) -> np.float32:
    """Sequential float sum (Eigen's reduction of a 2..4-element float vector)."""
    s = np.float32(0.0)
    for x in v:
        s = np.float32(s + np.float32(x))
    return s


def _libm_logf(x: float) -> float:
    return _libm.logf(x)


class MleTransitionUpdateConfig:
    """csrc/transition-model.h:80-92."""

    def __init__(self, floor: float = 0.01, mincount: float = 5.0, share_for_pdfs: bool = False):
        self.floor, self.mincount, self.share_for_pdfs = floor, mincount, share_for_pdfs


class TransitionModelTuple:
    def __init__(self, phone=0, hmm_state=0, forward_pdf=0, self_loop_pdf=0):
        self.phone, self.hmm_state, self.forward_pdf, self.self_loop_pdf = phone, hmm_state, forward_pdf, self_loop_pdf

    def _key(self):
        return (self.phone, self.hmm_state, self.forward_pdf, self.self_loop_pdf)

    def __eq__(self, o):
        return self._key() == o._key()

    def __lt__(self, o):
        return self._key() < o._key()

    def __str__(self):
        return (f"TransitionModelTuple(phone={self.phone},hmm_state={self.hmm_state},"
                f"forward_pdf={self.forward_pdf},self_loop_pdf={self.self_loop_pdf})")

    def __getstate__(self):
        return self._key()

    def __setstate__(self, t):
        self.phone, self.hmm_state, self.forward_pdf, self.self_loop_pdf = t


class TransitionModel:
    def __init__(self, ctx_dep=None, hmm_topo: HmmTopology = None):
        self._tuples: List[TransitionModelTuple] = []
        self._topo = hmm_topo
        self._state2id: List[int] = []
        self._id2state: List[int] = []
        self._id2pdf: List[int] = []
        self._num_pdfs = 0
        self._log_probs = np.zeros(0, np.float32)
        self._nsl = np.zeros(0, np.float32)
        if ctx_dep is not None:
            self._compute_tuples(ctx_dep)      # csrc/transition-model.cc:120-252
            self._compute_derived()            # :254-303
            self._initialize_probs()           # :318-337
            self.check()                       # :396-419

    # ---- construction -------------------------------------------------------------------
    def _compute_tuples(self, ctx_dep):
        topo = self._topo
        phones = topo.phones
        if not topo.is_hmm:
            raise KhgError("TransitionModel: only is_hmm topologies (PdfClass) are supported by the monophone tree")
        num_pdf_classes = [-1] * (max(phones) + 1)
        for ph in phones:
            num_pdf_classes[ph] = topo.num_pdf_classes(ph)
        pdf_info = ctx_dep.get_pdf_info(phones, num_pdf_classes)
        to_hmm_state = {}
        for ph in phones:
            for j, st in enumerate(topo.topology_for_phone(ph)):
                if st.forward_pdf_class != kNoPdf:
                    to_hmm_state.setdefault((ph, st.forward_pdf_class), []).append(j)
        tuples = []
        for pdf, lst in enumerate(pdf_info):
            for ph, pdf_class in lst:
                states = to_hmm_state.get((ph, pdf_class), [])
                if not states:
                    raise KhgError("ComputeTuplesIsHmm: no HMM state emits this pdf-class")
                for hs in states:
                    tuples.append(TransitionModelTuple(ph, hs, pdf, pdf))
        tuples.sort()
        self._tuples = tuples

    def _compute_derived(self):
        topo, tuples = self._topo, self._tuples
        n = len(tuples)
        self._state2id = [0] * (n + 2)
        cur = 1
        self._num_pdfs = 0
        for ts in range(1, n + 2):
            self._state2id[ts] = cur
            if ts <= n:
                t = tuples[ts - 1]
                self._num_pdfs = max(self._num_pdfs, 1 + t.forward_pdf, 1 + t.self_loop_pdf)
                cur += len(topo.topology_for_phone(t.phone)[t.hmm_state].transitions)
        self._id2state = [0] * cur
        self._id2pdf = [0] * cur
        for ts in range(1, n + 1):
            for tid in range(self._state2id[ts], self._state2id[ts + 1]):
                self._id2state[tid] = ts
                t = tuples[ts - 1]
                self._id2pdf[tid] = t.self_loop_pdf if self._is_self_loop_raw(tid) else t.forward_pdf

    def _is_self_loop_raw(self, tid):
        ts = self._id2state[tid]
        idx = tid - self._state2id[ts]
        t = self._tuples[ts - 1]
        tr = self._topo.topology_for_phone(t.phone)[t.hmm_state].transitions
        return idx < len(tr) and tr[idx][0] == t.hmm_state

    def _initialize_probs(self):
        nt = self.num_transition_ids
        lp = np.zeros(nt + 1, np.float32)
        for tid in range(1, nt + 1):
            ts = self._id2state[tid]
            idx = tid - self._state2id[ts]
            t = self._tuples[ts - 1]
            prob = np.float32(self._topo.topology_for_phone(t.phone)[t.hmm_state].transitions[idx][1])
            if prob <= 0.0:
                raise KhgError("TransitionModel::InitializeProbs, zero probability [should remove that entry in the topology]")
            lp[tid] = np.log(prob)
        self._log_probs = lp
        self._compute_derived_of_probs()

    def _compute_derived_of_probs(self):  # csrc/transition-model.cc:339-359
        n = self.num_transition_states
        nsl = np.zeros(n + 1, np.float32)
        for ts in range(1, n + 1):
            tid = self.self_loop_of(ts)
            if tid == 0:
                nsl[ts] = 0.0
            else:
                # libm's float expf / logf, exactly what the C++ update (khg_transition_mle_update) calls
                slp = _libm_expf(float(self._log_probs[tid]))
                p = np.float32(1.0 - float(slp))
                if p <= 0.0:
                    p = np.float32(1.0e-10)
                nsl[ts] = _libm_logf(float(p))
        self._nsl = nsl

    def check(self):
        if self.num_transition_ids == 0 or self.num_transition_states == 0:
            raise KhgError("TransitionModel::Check failed")
        for tid in range(1, self.num_transition_ids + 1):
            lp = self._log_probs[tid]
            if not (lp <= 0.0 and lp - lp == 0.0):
                raise KhgError("TransitionModel::Check: bad log prob")

    # ---- TransitionInformation (csrc/transition-information.h) ----
    @property
    def num_transition_ids(self) -> int:
        return len(self._id2state) - 1

    @property
    def num_transition_states(self) -> int:
        return len(self._tuples)

    @property
    def num_pdfs(self) -> int:
        return self._num_pdfs

    @property
    def topo(self):
        return self._topo

    @property
    def phones(self):
        return self._topo.phones

    @property
    def tuples(self):
        return self._tuples

    @property
    def state2id(self):
        return list(self._state2id)

    @property
    def id2state(self):
        return list(self._id2state)

    @property
    def id2pdf_id(self):
        return list(self._id2pdf)

    @property
    def log_probs(self):
        return self._log_probs.tolist()

    @property
    def non_self_loop_log_probs(self):
        return self._nsl.tolist()

    def _chk(self, tid):
        if not (0 < tid <= self.num_transition_ids):
            raise KhgError(f"transition-id {tid} out of range")

    def transition_id_to_pdf(self, trans_id: int) -> int:
        self._chk(trans_id)   # the reference reads INT_MAX here without a bounds check (Appendix A-8)
        return self._id2pdf[trans_id]

    def transition_id_to_pdf_array(self):
        return list(self._id2pdf)

    def transition_id_to_phone(self, trans_id):
        self._chk(trans_id)
        return self._tuples[self._id2state[trans_id] - 1].phone

    def transition_id_to_hmm_state(self, trans_id):
        self._chk(trans_id)
        return self._tuples[self._id2state[trans_id] - 1].hmm_state

    def transition_ids_equivalent(self, a, b):
        self._chk(a); self._chk(b)
        return self._id2state[a] == self._id2state[b]

    def transition_ids_is_start_of_phone(self, trans_id):
        return self.transition_id_to_hmm_state(trans_id) == 0

    def is_self_loop(self, trans_id):
        self._chk(trans_id)
        return self._is_self_loop_raw(trans_id)

    def is_final(self, trans_id):
        self._chk(trans_id)
        ts = self._id2state[trans_id]
        idx = trans_id - self._state2id[ts]
        t = self._tuples[ts - 1]
        entry = self._topo.topology_for_phone(t.phone)
        return entry[t.hmm_state].transitions[idx][0] + 1 == len(entry)

    def self_loop_of(self, trans_state: int) -> int:
        t = self._tuples[trans_state - 1]
        for idx, (dst, _) in enumerate(self._topo.topology_for_phone(t.phone)[t.hmm_state].transitions):
            if dst == t.hmm_state:
                return self._state2id[trans_state] + idx
        return 0

    def get_transition_log_prob(self, trans_id):
        return float(self._log_probs[trans_id])

    def tuple_to_transition_state(self, phone: int, hmm_state: int, pdf: int, self_loop_pdf: int) -> int:
        """csrc/transition-model.cc:432-447 (1-based; throws when the tuple is absent)."""
        import bisect
        t = TransitionModelTuple(phone, hmm_state, pdf, self_loop_pdf)
        i = bisect.bisect_left(self._tuples, t)
        if i == len(self._tuples) or not (self._tuples[i] == t):
            raise KhgError("TransitionModel::TupleToTransitionState, tuple not found. (incompatible tree and model?)")
        return i + 1

    def pair_to_transition_id(self, trans_state: int, trans_index: int) -> int:
        """csrc/transition-model.cc:385-390"""
        if not (0 < trans_state <= len(self._tuples)) or not (0 <= trans_index < self._state2id[trans_state + 1] - self._state2id[trans_state]):
            raise KhgError("PairToTransitionId: out of range")
        return self._state2id[trans_state] + trans_index

    def transition_id_to_transition_state(self, trans_id: int) -> int:
        self._chk(trans_id)
        return self._id2state[trans_id]

    def get_non_self_loop_log_prob(self, trans_state: int) -> float:
        """csrc/transition-model.cc:515-518"""
        return float(self._nsl[trans_state])

    def get_transition_log_prob_ignoring_self_loops(self, trans_id: int) -> float:
        """csrc/transition-model.cc:520-526 (float32 subtraction like the reference)"""
        if self.is_self_loop(trans_id):
            raise KhgError("GetTransitionLogProbIgnoringSelfLoops: self-loop")
        return float(np.float32(self._log_probs[trans_id]) - np.float32(self._nsl[self._id2state[trans_id]]))

    # ---- statistics (csrc/transition-model.h:176-189) ----
    def init_stats(self) -> np.ndarray:
        return np.zeros(self.num_transition_ids + 1, np.float64)

    def accumulate(self, prob: float, trans_id: int, stats: np.ndarray) -> np.ndarray:
        self._chk(trans_id)
        stats = np.asarray(stats, np.float64)
        stats[trans_id] += prob
        return stats

    def mle_update(self, stats, cfg: MleTransitionUpdateConfig = None):
        """csrc/transition-model.cc:657-750 -> (objf_impr, count)."""
        cfg = cfg or MleTransitionUpdateConfig()
        st = _lib.as_np(np.asarray(stats), np.float64)
        if st.shape[0] != self.num_transition_ids + 1:
            raise KhgError("stats.size() == NumTransitionIds() + 1 assertion failed")
        if cfg.share_for_pdfs:
            return self._mle_update_shared(st, cfg)
        s2i = np.asarray(self._state2id, np.int32)
        slo = np.asarray([0] + [self.self_loop_of(ts) for ts in range(1, self.num_transition_states + 1)], np.int32)
        oi, cnt = C.c_float(), C.c_float()
        check(lib.khg_transition_mle_update(self.num_transition_states, ptr(s2i, C.c_int32), ptr(slo, C.c_int32),
                                            ptr(st, C.c_double), C.c_float(cfg.floor), C.c_float(cfg.mincount),
                                            ptr(self._log_probs, C.c_float), ptr(self._nsl, C.c_float),
                                            C.byref(oi), C.byref(cnt)))
        return oi.value, cnt.value

    def _mle_update_shared(self, st: np.ndarray, cfg: "MleTransitionUpdateConfig"):
        """TransitionModel::MleUpdateShared (csrc/transition-model.cc:531-655): one set of transition probabilities for all
        transition-states that share a pdf.  Arithmetic as there: counts and their total in double, the new probabilities a
        float vector (normalised and floored three times), the objective change summed in float."""
        f32 = np.float32
        groups = {}                                   # pdf -> ordered set of transition-states (std::map<int32, std::set<int32>>)
        for ts in range(1, self.num_transition_states + 1):
            t = self._tuples[ts - 1]
            groups.setdefault(t.forward_pdf, set()).add(ts)
            if not self._topo.is_hmm:
                groups.setdefault(t.self_loop_pdf, set()).add(ts)
        count_sum, objf_sum = f32(0.0), f32(0.0)
        floor = f32(cfg.floor)
        for pdf in sorted(groups):
            tstates = sorted(groups[pdf])
            one = tstates[0]
            n = self._state2id[one + 1] - self._state2id[one]
            if n <= 1:
                continue
            counts = np.zeros(n, np.float64)
            pdf_tot = 0.0
            for ts in tstates:
                if self._state2id[ts + 1] - self._state2id[ts] != n:
                    raise KhgError("Mismatch in #transition indices: you cannot use the --share-for-pdfs option with this topology "
                                   "and sharing scheme.")
                for k in range(n):
                    acc = float(st[self._state2id[ts] + k])
                    counts[k] += acc
                    pdf_tot += acc
            count_sum = f32(np.float64(count_sum) + pdf_tot)          # float += double
            if pdf_tot < cfg.mincount:
                continue
            old = np.array([_libm_expf(float(self._log_probs[self._state2id[one] + k])) for k in range(n)], f32)   # GetTransitionProb
            new = (counts / pdf_tot).astype(f32)
            for _ in range(3):                          # keep flooring + renormalising three times
                new = (new / _f32_sum(new)).astype(f32)
                new = np.maximum(new, floor)
            for k in range(n):
                dlog = f32(f32(_libm_logf(float(new[k]))) - f32(_libm_logf(float(old[k]))))       # std::log(float) is logf
                objf_sum = f32(np.float64(objf_sum) + counts[k] * np.float64(dlog))                # float += double
            for ts in tstates:
                for k in range(n):
                    lp = _libm_logf(float(new[k]))
                    if not np.isfinite(lp):
                        raise KhgError("Log probs is inf or NaN: error in update or bad stats?")
                    self._log_probs[self._state2id[ts] + k] = lp
        self._compute_derived_of_probs()
        return float(objf_sum), float(count_sum)

    # ---- helpers for the device path ----
    def is_self_loop_array(self) -> np.ndarray:
        a = np.zeros(self.num_transition_ids + 1, np.uint8)
        for tid in range(1, self.num_transition_ids + 1):
            a[tid] = self._is_self_loop_raw(tid)
        return a

    def scaled_trans_cost(self, transition_scale: float, self_loop_scale: float) -> np.ndarray:
        """-GetScaledTransitionLogProb for every tid (csrc/hmm-utils.cc:442-463)."""
        out = np.zeros(self.num_transition_ids + 1, np.float32)
        i2s = np.asarray(self._id2state, np.int32)
        sl = self.is_self_loop_array()
        check(lib.khg_scaled_trans_cost(self.num_transition_ids, ptr(self._log_probs, C.c_float), ptr(self._nsl, C.c_float),
                                        ptr(i2s, C.c_int32), ptr(sl, C.c_uint8), C.c_float(transition_scale),
                                        C.c_float(self_loop_scale), ptr(out, C.c_float)))
        return out

    # ---- stream I/O (csrc/transition-model.cc:37-116), text and binary ----
    def _write(self, w) -> None:
        if not w.binary:
            w.raw(str(self))
            return
        hmm = self._topo.is_hmm
        w.token("<TransitionModel>")
        self._topo._write(w)
        w.token("<Triples>" if hmm else "<Tuples>")
        w.int32(len(self._tuples))
        for t in self._tuples:
            w.int32(t.phone); w.int32(t.hmm_state); w.int32(t.forward_pdf)
            if not hmm:
                w.int32(t.self_loop_pdf)
        w.token("</Triples>" if hmm else "</Tuples>")
        w.token("<LogProbs>")
        w.float_vector(self._log_probs)
        w.token("</LogProbs>")
        w.token("</TransitionModel>")

    def _read(self, r) -> None:
        r.expect("<TransitionModel>")
        self._topo = HmmTopology()
        self._topo._read(r)
        tok = r.token()
        if tok not in ("<Triples>", "<Tuples>"):
            raise KhgError(f"TransitionModel::Read, unexpected token {tok}")
        n = r.int32()
        self._tuples = []
        for _ in range(n):
            ph, hs, fp = r.int32(), r.int32(), r.int32()
            self._tuples.append(TransitionModelTuple(ph, hs, fp, r.int32() if tok == "<Tuples>" else fp))
        end = r.token()
        if end not in ("</Triples>", "</Tuples>"):
            raise KhgError(f"TransitionModel::Read, unexpected token {end}")
        self._compute_derived()
        r.expect("<LogProbs>")
        self._log_probs = np.asarray(r.float_vector(), np.float32).copy()
        r.expect("</LogProbs>")
        r.expect("</TransitionModel>")
        if self._log_probs.shape[0] != self.num_transition_ids + 1:
            raise KhgError("TransitionModel::Read: <LogProbs> size does not match the tuples")
        self._compute_derived_of_probs()
        self.check()

    def write(self, binary: bool, filename: str) -> None:
        from . import kaldi_io
        w = kaldi_io.Writer(binary)
        self._write(w)
        kaldi_io.write_file(filename, binary, w.getvalue())

    def read(self, filename: str) -> None:
        from . import kaldi_io
        self._read(kaldi_io.read_file(filename))

    def __str__(self):  # csrc/transition-model.cc:37-83 text Write
        out = ["<TransitionModel> \n", str(self._topo), "<Triples> ", f"{len(self._tuples)} \n"]
        for t in self._tuples:
            out.append(f"{t.phone} {t.hmm_state} {t.forward_pdf} \n")
        out.append("</Triples> \n<LogProbs> \n [ ")
        out.append(" ".join("%g" % x for x in self._log_probs))
        out.append(" ]\n</LogProbs> \n</TransitionModel> \n")
        return "".join(out)

    # pickle: 8-tuple, python/csrc/transition-model.cc:122-150
    def __getstate__(self):
        return (self._tuples, self._topo, self._state2id, self._id2state, self._id2pdf, self._num_pdfs,
                self._log_probs.tolist(), self._nsl.tolist())

    def __setstate__(self, t):
        (self._tuples, self._topo, self._state2id, self._id2state, self._id2pdf, self._num_pdfs) = t[:6]
        self._log_probs = np.asarray(t[6], np.float32)
        self._nsl = np.asarray(t[7], np.float32)


def get_pdfs_for_phones(trans_model: TransitionModel, phones: List[int]):
    """csrc/transition-model.cc:752-785 -> (is_unique, pdfs)."""
    if sorted(set(phones)) != list(phones):
        raise KhgError("IsSortedAndUniq(phones) assertion failed")
    ps = set(phones)
    pdfs = set()
    for t in trans_model.tuples:
        if t.phone in ps:
            pdfs.add(t.forward_pdf)
            pdfs.add(t.self_loop_pdf)
    ok = True
    for t in trans_model.tuples:
        if (t.forward_pdf in pdfs or t.self_loop_pdf in pdfs) and t.phone not in ps:
            ok = False
    return ok, sorted(pdfs)
