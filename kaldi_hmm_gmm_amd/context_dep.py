"""ContextDependency over Kaldi EventMaps (csrc/event-map.{h,cc}, csrc/context-dep.{h,cc}).

The tree object of the reference: an EventMap from {(-1 = kPdfClass, pdf_class), (i, phone at window
position i)} to a pdf-id.  ``monophone_context_dependency[_shared]`` build the stub trees of
csrc/build-tree-utils.cc:18-121 (GetStubMap) with the reference's node structure, so ``write`` emits
the same text/binary files; ``read`` loads any Kaldi ``tree`` file (ConstantEventMap "CE",
TableEventMap "TE", SplitEventMap "SE"), including context-dependent ones.
"""
from typing import Dict, List, Optional, Sequence, Tuple

from . import kaldi_io
from ._lib import KhgError

kPdfClass = -1   # csrc/event-map.h: the key of the pdf-class in an event vector


class EventMap:
    def map(self, event: Dict[int, int]) -> Optional[int]:
        raise NotImplementedError

    def multi_map(self, event: Dict[int, int], ans: List[int]) -> None:
        raise NotImplementedError

    def write(self, w: kaldi_io.Writer) -> None:
        raise NotImplementedError

    def max_result(self) -> int:
        ans: List[int] = []
        self.multi_map({}, ans)
        return max(ans) if ans else -2**31      # csrc/event-map.h:152-164

    @staticmethod
    def write_any(w: kaldi_io.Writer, m: Optional["EventMap"]) -> None:   # csrc/event-map.cc:116-122
        if m is None:
            w.token("NULL")
        else:
            m.write(w)

    @staticmethod
    def read(r: kaldi_io.Reader) -> Optional["EventMap"]:                  # csrc/event-map.cc:124-140
        c = r.peek()
        if c == "N":
            r.expect("NULL")
            return None
        if c == "C":
            return ConstantEventMap._read(r)
        if c == "T":
            return TableEventMap._read(r)
        if c == "S":
            return SplitEventMap._read(r)
        raise KhgError(f"EventMap::read, was not expecting character {c!r}")


class ConstantEventMap(EventMap):
    def __init__(self, answer: int):
        self.answer = int(answer)

    def map(self, event):
        return self.answer

    def multi_map(self, event, ans):
        ans.append(self.answer)

    def write(self, w):                       # csrc/event-map.cc:142-148
        w.token("CE")
        w.int32(self.answer)

    @staticmethod
    def _read(r):
        r.expect("CE")
        return ConstantEventMap(r.int32())


class TableEventMap(EventMap):
    def __init__(self, key: int, table: Sequence[Optional[EventMap]]):
        self.key = int(key)
        self.table = list(table)

    @staticmethod
    def from_map(key: int, m: Dict[int, EventMap]) -> "TableEventMap":   # csrc/event-map.cc:266-300
        if not m:
            return TableEventMap(key, [])
        table: List[Optional[EventMap]] = [None] * (max(m) + 1)
        for v, e in m.items():
            if v < 0:
                raise KhgError("TableEventMap: negative value")
            table[v] = e
        return TableEventMap(key, table)

    def map(self, event):                     # csrc/event-map.h:227-237
        v = event.get(self.key)
        if v is not None and 0 <= v < len(self.table) and self.table[v] is not None:
            return self.table[v].map(event)
        return None

    def multi_map(self, event, ans):          # csrc/event-map.h:248-264
        v = event.get(self.key)
        if v is not None:
            if 0 <= v < len(self.table) and self.table[v] is not None:
                self.table[v].multi_map(event, ans)
        else:
            for e in self.table:
                if e is not None:
                    e.multi_map(event, ans)

    def write(self, w):                       # csrc/event-map.cc:216-237
        w.token("TE")
        w.int32(self.key)
        w.uint32(len(self.table))
        w.token("(")
        for e in self.table:
            EventMap.write_any(w, e)
        w.token(")")
        w.nl()

    @staticmethod
    def _read(r):
        r.expect("TE")
        key = r.int32()
        n = r.uint32()
        r.expect("(")
        table = [EventMap.read(r) for _ in range(n)]
        r.expect(")")
        return TableEventMap(key, table)


class SplitEventMap(EventMap):
    def __init__(self, key: int, yes_set: Sequence[int], yes: EventMap, no: EventMap):
        if yes is None or no is None:
            raise KhgError("SplitEventMap: NULL children are not valid")
        self.key = int(key)
        self.yes_set = sorted(set(int(x) for x in yes_set))
        self._yes_lookup = set(self.yes_set)
        self.yes, self.no = yes, no

    def map(self, event):                     # csrc/event-map.h:307-317
        v = event.get(self.key)
        if v is None:
            return None
        return (self.yes if v in self._yes_lookup else self.no).map(event)

    def multi_map(self, event, ans):          # csrc/event-map.h:319-332
        v = event.get(self.key)
        if v is not None:
            (self.yes if v in self._yes_lookup else self.no).multi_map(event, ans)
        else:
            self.yes.multi_map(event, ans)
            self.no.multi_map(event, ans)

    def write(self, w):                       # csrc/event-map.cc:334-352
        w.token("SE")
        w.int32(self.key)
        w.int_vector(self.yes_set)
        w.token("{")
        self.yes.write(w)
        self.no.write(w)
        w.token("}")
        w.nl()

    @staticmethod
    def _read(r):
        r.expect("SE")
        key = r.int32()
        ys = r.int_vector()
        r.expect("{")
        yes = EventMap.read(r)
        no = EventMap.read(r)
        r.expect("}")
        if yes is None or no is None:
            raise KhgError("SplitEventMap::Read, NULL pointers.")
        return SplitEventMap(key, ys, yes, no)


def get_stub_map(P: int, phone_sets: List[List[int]], phone2num_pdf_classes: List[int], share_roots: List[bool],
                 num_leaves: List[int]) -> EventMap:
    """csrc/build-tree-utils.cc:18-121 (num_leaves is a one-element in/out counter)."""
    if not phone_sets or len(share_roots) != len(phone_sets):
        raise KhgError("GetStubMap: bad arguments")
    seen = set()
    for ps in phone_sets:
        if not ps or sorted(set(ps)) != list(ps):
            raise KhgError("GetStubMap: phone sets must be non-empty, sorted and unique")
        for p in ps:
            if p in seen:
                raise KhgError("GetStubMap: phone appears in two sets")
            seen.add(p)
    max_set_size = max(len(ps) for ps in phone_sets)
    highest = max(max(ps) for ps in phone_sets)
    if len(phone_sets) == 1:
        if share_roots[0]:
            num_leaves[0] += 1
            return ConstantEventMap(num_leaves[0] - 1)
        max_len = 0
        for i, phone in enumerate(phone_sets[0]):
            if phone >= len(phone2num_pdf_classes) or phone2num_pdf_classes[phone] <= 0:
                raise KhgError("GetStubMap: phone without pdf classes")
            max_len = max(max_len, phone2num_pdf_classes[phone])      # mismatching lengths: warn + max
        m = {}
        for pos in range(max_len):
            m[pos] = ConstantEventMap(num_leaves[0])
            num_leaves[0] += 1
        return TableEventMap.from_map(kPdfClass, m)
    if max_set_size == 1 and len(phone_sets) <= 2 * highest:
        m = {}
        for ps, sr in zip(phone_sets, share_roots):
            m[ps[0]] = get_stub_map(P, [ps], phone2num_pdf_classes, [sr], num_leaves)
        return TableEventMap.from_map(P, m)
    half = len(phone_sets) // 2
    map1 = get_stub_map(P, phone_sets[:half], phone2num_pdf_classes, share_roots[:half], num_leaves)
    map2 = get_stub_map(P, phone_sets[half:], phone2num_pdf_classes, share_roots[half:], num_leaves)
    first = sorted(p for ps in phone_sets[:half] for p in ps)
    return SplitEventMap(P, first, map1, map2)


class ContextDependencyInterface:
    """csrc/context-dep.h (ContextDependencyInterface; python/csrc/context-dep.cc:17-53): what TransitionModel and the graph
    compiler ask of a tree."""

    @property
    def context_width(self) -> int:
        raise NotImplementedError

    @property
    def central_position(self) -> int:
        raise NotImplementedError

    @property
    def num_pdfs(self) -> int:
        raise NotImplementedError

    def compute(self, phone_seq: List[int], pdf_class: int):
        raise NotImplementedError

    def get_pdf_info(self, phones: List[int], num_pdf_classes: List[int]) -> List[List[Tuple[int, int]]]:
        raise NotImplementedError


class ContextDependency(ContextDependencyInterface):
    """csrc/context-dep.h: (N, P, to_pdf)."""

    def __init__(self, N: int = 1, P: int = 0, to_pdf: Optional[EventMap] = None):
        self._N, self._P, self._to_pdf = int(N), int(P), to_pdf

    @property
    def context_width(self) -> int:
        return self._N

    @property
    def central_position(self) -> int:
        return self._P

    @property
    def num_pdfs(self) -> int:             # csrc/context-dep.cc: to_pdf_->MaxResult() + 1
        if self._to_pdf is None:
            return 0
        return self._to_pdf.max_result() + 1

    @property
    def to_pdf(self) -> EventMap:
        return self._to_pdf

    def compute(self, phone_seq: List[int], pdf_class: int):
        """csrc/context-dep.cc:22-43 -> (ok, pdf_id)   (argument names of python/csrc/context-dep.cc:22-31)."""
        seq = phone_seq
        if len(seq) != self._N:
            raise KhgError(f"ContextDependency::Compute: expected {self._N} phones, got {len(seq)}")
        event = {kPdfClass: int(pdf_class)}
        for i, p in enumerate(seq):
            event[i] = int(p)
        ans = self._to_pdf.map(event)
        if ans is None:
            return False, -1
        return True, ans

    def get_pdf_info(self, phones: List[int], num_pdf_classes: List[int]) -> List[List[Tuple[int, int]]]:
        """csrc/context-dep.cc:165-205: pdf -> sorted [(phone, pdf_class)]."""
        info: List[List[Tuple[int, int]]] = [[] for _ in range(self.num_pdfs)]
        for ph in phones:
            for pos in range(num_pdf_classes[ph]):
                pdfs: List[int] = []
                self._to_pdf.multi_map({self._P: ph, kPdfClass: pos}, pdfs)
                for pdf in sorted(set(pdfs)):
                    info[pdf].append((ph, pos))
        return [sorted(x) for x in info]

    # ---- I/O (csrc/context-dep.cc:45-83) ----
    def _write(self, w: kaldi_io.Writer):
        w.token("ContextDependency")
        w.int32(self._N)
        w.int32(self._P)
        w.token("ToPdf")
        self._to_pdf.write(w)
        w.token("EndContextDependency")

    def to_bytes(self, binary: bool) -> bytes:
        w = kaldi_io.Writer(binary)
        self._write(w)
        return w.getvalue()

    def write(self, binary: bool, filename: str):
        kaldi_io.write_file(filename, binary, self.to_bytes(binary))

    def _read(self, r: kaldi_io.Reader):
        r.expect("ContextDependency")
        self._N = r.int32()
        self._P = r.int32()
        tok = r.token()
        if tok == "ToLength":                 # back-compat
            EventMap.read(r)
            tok = r.token()
        if tok != "ToPdf":
            raise KhgError(f"Got unexpected token {tok} reading context-dependency object.")
        self._to_pdf = EventMap.read(r)
        r.expect("EndContextDependency")

    def read(self, filename: str):
        self._read(kaldi_io.read_file(filename))

    @staticmethod
    def from_bytes(data: bytes, binary: bool) -> "ContextDependency":
        c = ContextDependency()
        c._read(kaldi_io.Reader(data, binary))
        return c

    def __getstate__(self):                   # python/csrc/context-dep.cc:64-79: the binary Write as the pickle state
        return self.to_bytes(True)

    def __setstate__(self, data):
        self._read(kaldi_io.Reader(data, True))

    def __str__(self):                        # python/csrc/context-dep.cc:58-62: text Write
        return self.to_bytes(False).decode("ascii")


def monophone_context_dependency(phones: List[int], phone2num_pdf_classes: List[int]) -> ContextDependency:
    """csrc/context-dep.cc:241-255"""
    n = [0]
    m = get_stub_map(0, [[p] for p in phones], list(phone2num_pdf_classes), [False] * len(phones), n)
    return ContextDependency(1, 0, m)


def monophone_context_dependency_shared(phone_classes: List[List[int]], phone2num_pdf_classes: List[int]) -> ContextDependency:
    """csrc/context-dep.cc:257-268"""
    n = [0]
    m = get_stub_map(0, [list(ps) for ps in phone_classes], list(phone2num_pdf_classes), [False] * len(phone_classes), n)
    return ContextDependency(1, 0, m)
