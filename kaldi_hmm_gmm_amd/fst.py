"""Minimal StdVectorFst (tropical semiring) with the kaldifst method names the reference's scripts
use (scripts/gmm_align_compiled.py:6-41, egs/yesno/train.py:70-108): the `kaldifst` package that
registers fst::VectorFst<StdArc> for the reference is not available offline, so the decoding-graph
argument of this API is this class.  Final weight +inf == TropicalWeight::Zero() (not final)."""
import math
from typing import List

import numpy as np

from ._lib import KhgError

kNoStateId = -1


class StdArc:
    __slots__ = ("ilabel", "olabel", "weight", "nextstate")

    def __init__(self, ilabel: int, olabel: int, weight: float, nextstate: int):
        self.ilabel, self.olabel, self.weight, self.nextstate = int(ilabel), int(olabel), float(np.float32(weight)), int(nextstate)

    def __repr__(self):
        return f"StdArc({self.ilabel}, {self.olabel}, {self.weight:g}, {self.nextstate})"


class StdVectorFst:
    def __init__(self):
        self._arcs: List[List[StdArc]] = []
        self._final: List[float] = []
        self._start = kNoStateId

    def add_state(self) -> int:
        self._arcs.append([])
        self._final.append(math.inf)
        return len(self._arcs) - 1

    @property
    def num_states(self) -> int:
        return len(self._arcs)

    @property
    def start(self) -> int:
        return self._start

    @start.setter
    def start(self, s: int):
        self._start = int(s)

    def set_start(self, s: int):
        self._start = int(s)

    def add_arc(self, state: int, arc: StdArc = None, **kw):
        if arc is None:
            arc = StdArc(kw["ilabel"], kw["olabel"], kw.get("weight", 0.0), kw["nextstate"])
        if not 0 <= state < len(self._arcs):
            raise KhgError("add_arc: bad state")
        self._arcs[state].append(arc)

    def set_final(self, state: int, weight: float = 0.0):
        self._final[state] = float(np.float32(weight))

    def final(self, state: int) -> float:
        return self._final[state]

    def is_final(self, state: int) -> bool:
        return self._final[state] != math.inf

    def arcs(self, state: int) -> List[StdArc]:
        return self._arcs[state]

    def num_arcs(self, state: int = None) -> int:
        return len(self._arcs[state]) if state is not None else sum(len(a) for a in self._arcs)

    def copy(self) -> "StdVectorFst":
        f = StdVectorFst()
        f._arcs = [[StdArc(a.ilabel, a.olabel, a.weight, a.nextstate) for a in arcs] for arcs in self._arcs]
        f._final = list(self._final)
        f._start = self._start
        return f

    # ---- flat CSR used by the device path ----
    def to_csr(self):
        off = np.zeros(self.num_states + 1, np.int64)
        for s, arcs in enumerate(self._arcs):
            off[s + 1] = off[s] + len(arcs)
        n = int(off[-1])
        il = np.zeros(n, np.int32); ol = np.zeros(n, np.int32); w = np.zeros(n, np.float32); ns = np.zeros(n, np.int32)
        k = 0
        for arcs in self._arcs:
            for a in arcs:
                il[k], ol[k], w[k], ns[k] = a.ilabel, a.olabel, a.weight, a.nextstate
                k += 1
        return {"start": self._start, "arc_off": off, "ilabel": il, "olabel": ol, "weight": w, "nextstate": ns,
                "final": np.asarray(self._final, np.float32)}

    @staticmethod
    def from_csr(start, arc_off, ilabel, olabel, weight, nextstate, final) -> "StdVectorFst":
        f = StdVectorFst()
        for s in range(len(final)):
            f.add_state()
            if final[s] != math.inf:
                f._final[s] = float(final[s])
            for a in range(int(arc_off[s]), int(arc_off[s + 1])):
                f._arcs[s].append(StdArc(ilabel[a], olabel[a], weight[a], nextstate[a]))
        f._start = int(start)
        return f


def concat_graphs(fsts: List[StdVectorFst]) -> dict:
    """Concatenate per-utterance graphs into the CSR block khg_utts_create consumes."""
    state_off = [0]
    arc_off = [np.zeros(1, np.int64)]
    parts = {k: [] for k in ("ilabel", "olabel", "weight", "nextstate", "final")}
    start = []
    na = 0
    for f in fsts:
        c = f.to_csr()
        state_off.append(state_off[-1] + f.num_states)
        arc_off.append(c["arc_off"][1:] + na)
        na += int(c["arc_off"][-1])
        for k in parts:
            parts[k].append(c[k])
        start.append(c["start"])
    out = {k: (np.concatenate(v) if v else np.zeros(0)) for k, v in parts.items()}
    out["state_off"] = np.asarray(state_off, np.int64)
    out["arc_off"] = np.concatenate(arc_off)
    out["start"] = np.asarray(start, np.int32)
    return out


def modify_graph_for_careful_alignment(fst: StdVectorFst) -> None:
    """csrc/decoder-wrappers.cc:111-140 (+ OpenFst Concat): in place."""
    S = fst.num_states
    if S == 0:
        return
    rhs_arcs = [[StdArc(a.ilabel, a.olabel, a.weight, a.nextstate + S) for a in arcs] for arcs in fst._arcs]
    pre_initial = 2 * S
    for s in range(S):
        if fst._final[s] != math.inf:
            fst._arcs[s].append(StdArc(0, 0, fst._final[s], pre_initial))
            fst._final[s] = math.inf
    fst._arcs.extend(rhs_arcs)
    fst._final.extend([math.inf] * S)
    fst._arcs.append([StdArc(0, 0, 0.0, fst._start + S)])
    fst._final.append(0.0)
