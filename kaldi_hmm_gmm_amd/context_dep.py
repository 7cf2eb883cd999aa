"""Monophone ContextDependency -- the only tree the reference can build itself
(csrc/context-dep.cc:241-268 via GetStubMap, csrc/build-tree-utils.cc:18-121): context width 1,
central position 0; phone set i owns max_len_i consecutive pdf-ids, one per pdf-class."""
from typing import List, Tuple

from ._lib import KhgError


class ContextDependency:
    def __init__(self, phone_sets: List[List[int]], phone2num_pdf_classes: List[int]):
        seen = set()
        for ps in phone_sets:
            if not ps or sorted(set(ps)) != list(ps):
                raise KhgError("GetStubMap: phone sets must be non-empty, sorted and unique")
            for p in ps:
                if p in seen:
                    raise KhgError("GetStubMap: phone appears in two sets")
                seen.add(p)
        self._phone_sets = [list(ps) for ps in phone_sets]
        self._p2n = list(phone2num_pdf_classes)
        self._base = {}
        self._len = {}
        n = 0
        for ps in self._phone_sets:
            lens = []
            for p in ps:
                if p >= len(self._p2n) or self._p2n[p] <= 0:
                    raise KhgError("GetStubMap: phone without pdf classes")
                lens.append(self._p2n[p])
            for p in ps:
                self._base[p] = n
                self._len[p] = max(lens)
            n += max(lens)
        self._num_pdfs = n

    context_width = 1
    central_position = 0

    @property
    def num_pdfs(self) -> int:
        return self._num_pdfs

    def compute(self, phoneseq: List[int], pdf_class: int):
        """csrc/context-dep.cc:22-43 -> (ok, pdf_id)."""
        if len(phoneseq) != 1:
            raise KhgError("ContextDependency::Compute: context width is 1")
        p = phoneseq[0]
        if p not in self._base or not (0 <= pdf_class < self._len[p]):
            return False, -1
        return True, self._base[p] + pdf_class

    def get_pdf_info(self, phones: List[int], num_pdf_classes: List[int]) -> List[List[Tuple[int, int]]]:
        """csrc/context-dep.cc:165-205: pdf -> sorted [(phone, pdf_class)]."""
        info = [[] for _ in range(self._num_pdfs)]
        for ph in phones:
            for pos in range(num_pdf_classes[ph]):
                ok, pdf = self.compute([ph], pos)
                if ok:
                    info[pdf].append((ph, pos))
        return [sorted(x) for x in info]

    def __getstate__(self):
        return (self._phone_sets, self._p2n)

    def __setstate__(self, t):
        self.__init__(t[0], t[1])

    def __str__(self):
        return f"ContextDependency(N=1, P=0, num_pdfs={self._num_pdfs})"


def monophone_context_dependency(phones: List[int], phone2num_pdf_classes: List[int]) -> ContextDependency:
    return ContextDependency([[p] for p in phones], phone2num_pdf_classes)


def monophone_context_dependency_shared(phone_classes: List[List[int]], phone2num_pdf_classes: List[int]) -> ContextDependency:
    return ContextDependency(phone_classes, phone2num_pdf_classes)
