"""HmmTopology -- host-side mirror of the reference class (csrc/hmm-topology.{h,cc},
python/csrc/hmm-topology.cc): Kaldi text format, Check(), pickling as (phones, phone2idx, entries)."""
from typing import List, Tuple

from ._lib import KhgError

kNoPdf = -1


class HmmState:
    """csrc/hmm-topology.h HmmState: forward/self-loop pdf-class + [(dst_state, prob)]."""

    def __init__(self, forward_pdf_class: int = kNoPdf, self_loop_pdf_class: int = None, transitions=None):
        self.forward_pdf_class = forward_pdf_class
        self.self_loop_pdf_class = forward_pdf_class if self_loop_pdf_class is None else self_loop_pdf_class
        self.transitions: List[Tuple[int, float]] = list(transitions or [])

    def __eq__(self, o):
        return (self.forward_pdf_class, self.self_loop_pdf_class, self.transitions) == \
               (o.forward_pdf_class, o.self_loop_pdf_class, o.transitions)

    def __str__(self):  # python/csrc/hmm-topology.cc:22-36
        tr = ", ".join(f"({d}, {_fmt(p)})" for d, p in self.transitions)
        return (f"HmmState(forward_pdf_class={self.forward_pdf_class}, "
                f"self_loop_pdf_class={self.self_loop_pdf_class}, transitions=[{tr}])")

    def __getstate__(self):
        return (self.forward_pdf_class, self.self_loop_pdf_class, self.transitions)

    def __setstate__(self, t):
        self.forward_pdf_class, self.self_loop_pdf_class, self.transitions = t[0], t[1], list(t[2])


def _fmt(x: float) -> str:
    """C++ ostream << float (6 significant digits, %g-like)."""
    return "%g" % x


class HmmTopology:
    def __init__(self):
        self._phones: List[int] = []
        self._phone2idx: List[int] = []
        self._entries: List[List[HmmState]] = []

    # ---- csrc/hmm-topology.cc:23-160 (text mode) ----
    def read(self, s: str) -> None:
        tok = s.split()
        pos = 0

        def nxt():
            nonlocal pos
            if pos >= len(tok):
                raise KhgError("Reading HmmTopology object, unexpected end of input")
            pos += 1
            return tok[pos - 1]

        def expect(t):
            g = nxt()
            if g != t:
                raise KhgError(f"Expected token {t}, got {g}")

        expect("<Topology>")
        self._phones, self._phone2idx, self._entries = [], [], []
        while pos < len(tok):
            t = nxt()
            if t == "</Topology>":
                break
            if t != "<TopologyEntry>":
                raise KhgError("Reading HmmTopology object, expected </Topology> or <TopologyEntry>, got " + t)
            expect("<ForPhones>")
            phones = []
            while True:
                t = nxt()
                if t == "</ForPhones>":
                    break
                try:
                    phones.append(int(t))
                except ValueError:
                    raise KhgError("Reading HmmTopology object, expected integer, got instead " + t)
            entry: List[HmmState] = []
            t = nxt()
            while t != "</TopologyEntry>":
                if t != "<State>":
                    raise KhgError("Expected </TopologyEntry> or <State>, got instead " + t)
                state = int(nxt())
                if state != len(entry):
                    raise KhgError(f"States are expected to be in order from zero, expected {len(entry)}, got {state}")
                t = nxt()
                if t == "<PdfClass>":
                    entry.append(HmmState(int(nxt())))
                    t = nxt()
                    if t == "<SelfLoopPdfClass>":
                        raise KhgError("pdf classes should be defined using <PdfClass> or "
                                       "<ForwardPdfClass>/<SelfLoopPdfClass> pair")
                elif t == "<ForwardPdfClass>":
                    fwd = int(nxt())
                    t = nxt()
                    if t != "<SelfLoopPdfClass>":
                        raise KhgError("Expected <SelfLoopPdfClass>, got instead " + t)
                    entry.append(HmmState(fwd, int(nxt())))
                    t = nxt()
                else:
                    entry.append(HmmState(kNoPdf))
                while t == "<Transition>":
                    dst = int(nxt())
                    import numpy as np
                    prob = float(np.float32(float(nxt())))
                    entry[-1].transitions.append((dst, prob))
                    t = nxt()
                if t == "<Final>":
                    raise KhgError("You are trying to read old-format topology with new Kaldi.")
                if t != "</State>":
                    raise KhgError("Expected </State>, got instead " + t)
                t = nxt()
            idx = len(self._entries)
            self._entries.append(entry)
            for i, ph in enumerate(phones):
                if len(self._phone2idx) <= ph:
                    self._phone2idx.extend([-1] * (ph + 1 - len(self._phone2idx)))
                if ph <= 0:
                    raise KhgError("phone > 0 assertion failed")
                if self._phone2idx[ph] != -1:
                    raise KhgError(f"Phone with index {i} appears in multiple topology entries.")
                self._phone2idx[ph] = idx
                self._phones.append(ph)
        self._phones.sort()
        self.check()

    def __str__(self) -> str:  # csrc/hmm-topology.cc:162-218 (text mode)
        hmm = self.is_hmm
        out = ["<Topology> \n"]
        for i, entry in enumerate(self._entries):
            out.append("<TopologyEntry> \n<ForPhones> \n")
            out.append("".join(f"{j} " for j, k in enumerate(self._phone2idx) if k == i))
            out.append("\n</ForPhones> \n")
            for j, st in enumerate(entry):
                out.append(f"<State> {j} ")
                if st.forward_pdf_class != kNoPdf:
                    if hmm:
                        out.append(f"<PdfClass> {st.forward_pdf_class} ")
                    else:
                        out.append(f"<ForwardPdfClass> {st.forward_pdf_class} <SelfLoopPdfClass> {st.self_loop_pdf_class} ")
                for dst, p in st.transitions:
                    out.append(f"<Transition> {dst} {_fmt(p)} ")
                out.append("</State> \n")
            out.append("</TopologyEntry> \n")
        out.append("</Topology> \n")
        return "".join(out)

    # ---- stream I/O (csrc/hmm-topology.cc:23-282), text and binary ----
    def _write(self, w) -> None:
        if not w.binary:
            w.raw(str(self))
            return
        hmm = self.is_hmm
        w.token("<Topology>")
        w.int_vector(self._phones)
        w.int_vector(self._phone2idx)
        if not hmm:
            w.int32(-1)                 # marks the extended format with SelfLoopPdfClass
        w.int32(len(self._entries))
        for entry in self._entries:
            w.int32(len(entry))
            for st in entry:
                w.int32(st.forward_pdf_class)
                if not hmm:
                    w.int32(st.self_loop_pdf_class)
                w.int32(len(st.transitions))
                for dst, pr in st.transitions:
                    w.int32(dst)
                    w.float32(pr)
        w.token("</Topology>")

    def _read(self, r) -> None:
        if not r.binary:
            # text: hand the <Topology> ... </Topology> span to the token parser
            r._skip_ws()
            end = r.d.find(b"</Topology>", r.i)
            if end < 0:
                raise KhgError("Reading HmmTopology object, </Topology> not found")
            end += len(b"</Topology>")
            self.read(r.d[r.i:end].decode("ascii"))
            r.i = min(end + 1, len(r.d))
            return
        import numpy as np
        r.expect("<Topology>")
        self._phones = r.int_vector()
        self._phone2idx = r.int_vector()
        n = r.int32()
        hmm = True
        if n == -1:
            hmm = False
            n = r.int32()
        self._entries = []
        for _ in range(n):
            entry = []
            for _ in range(r.int32()):
                fwd = r.int32()
                st = HmmState(fwd, fwd if hmm else r.int32())
                for _ in range(r.int32()):
                    dst = r.int32()
                    st.transitions.append((dst, float(np.float32(r.float32()))))
                entry.append(st)
            self._entries.append(entry)
        r.expect("</Topology>")
        self.check()

    def write(self, binary: bool, filename: str) -> None:
        from . import kaldi_io
        w = kaldi_io.Writer(binary)
        self._write(w)
        kaldi_io.write_file(filename, binary, w.getvalue())

    def read_file(self, filename: str) -> None:
        from . import kaldi_io
        self._read(kaldi_io.read_file(filename))

    # ---- accessors ----
    @property
    def phones(self) -> List[int]:
        return list(self._phones)

    @property
    def is_hmm(self) -> bool:  # csrc/hmm-topology.cc:284-301
        return all(st.forward_pdf_class == st.self_loop_pdf_class for ph in self._phones
                   for st in self.topology_for_phone(ph))

    def topology_for_phone(self, phone: int) -> List[HmmState]:
        if phone >= len(self._phone2idx) or phone < 0 or self._phone2idx[phone] == -1:
            raise KhgError(f"TopologyForPhone(), phone {phone} not covered.")
        return self._entries[self._phone2idx[phone]]

    def num_pdf_classes(self, phone: int) -> int:
        m = 0
        for st in self.topology_for_phone(phone):
            m = max(m, st.forward_pdf_class, st.self_loop_pdf_class)
        return m + 1

    def get_phone_to_num_pdf_classes(self) -> List[int]:
        out = [-1] * (self._phones[-1] + 1)
        for ph in self._phones:
            out[ph] = self.num_pdf_classes(ph)
        return out

    def min_length(self, phone: int) -> int:  # csrc/hmm-topology.cc:453-492
        entry = self.topology_for_phone(phone)
        big = 2**31 - 1
        ml = [big] * len(entry)
        ml[0] = 0 if entry[0].forward_pdf_class == -1 else 1
        changed = True
        while changed:
            changed = False
            for s, st in enumerate(entry):
                for nxt_state, _ in st.transitions:
                    v = ml[s] + (0 if entry[nxt_state].forward_pdf_class == -1 else 1)
                    if ml[s] != big and v < ml[nxt_state]:
                        ml[nxt_state] = v
                        if nxt_state < s:
                            changed = True
        return ml[-1]

    def check(self) -> None:  # csrc/hmm-topology.cc:312-427
        if not self._entries or not self._phones or not self._phone2idx:
            raise KhgError("HmmTopology::Check(), empty object.")
        seen = [False] * len(self._entries)
        for ph in self._phones:
            if ph >= len(self._phone2idx) or not (0 <= self._phone2idx[ph] < len(self._entries)):
                raise KhgError("HmmTopology::Check(), phone has no valid index.")
            seen[self._phone2idx[ph]] = True
        for i, entry in enumerate(self._entries):
            if not seen[i]:
                raise KhgError("HmmTopoloy::Check(), entry with no corresponding phones.")
            n = len(entry)
            if n <= 1:
                raise KhgError("HmmTopology::Check(), cannot only have one state (i.e., must have at least one emitting state).")
            if entry[-1].transitions:
                raise KhgError("HmmTopology::Check(), last state must have no transitions.")
            if entry[-1].forward_pdf_class != kNoPdf:
                raise KhgError("HmmTopology::Check(), last state must not be emitting.")
            has_in = [False] * n
            classes = []
            for j, st in enumerate(entry):
                tot = 0.0
                if st.forward_pdf_class != kNoPdf:
                    classes += [st.forward_pdf_class, st.self_loop_pdf_class]
                seen_t = set()
                for dst, p in st.transitions:
                    tot += p
                    if p <= 0.0:
                        raise KhgError("HmmTopology::Check(), negative or zero transition prob.")
                    if dst == n - 1 and st.forward_pdf_class == kNoPdf:
                        raise KhgError("We do not allow any state to be nonemitting and have a transition to the final-state")
                    if dst < 0 or dst >= n:
                        raise KhgError(f"HmmTopology::Check(), invalid dest state {dst}")
                    if dst in seen_t:
                        raise KhgError("HmmTopology::Check(), duplicate transition found.")
                    seen_t.add(dst)
                    has_in[dst] = True
                if j + 1 < n:
                    if not tot > 0.0:
                        raise KhgError("Non-final state must have transitions out.(with nonzero probability)")
                elif tot != 0.0:
                    raise KhgError("assertion failed: tot_prob == 0.0")
            for j in range(1, n):
                if not has_in[j]:
                    raise KhgError(f"HmmTopology::Check, state {j} has no input transitions.")
            cs = sorted(set(classes))
            if cs[0] != 0 or cs[-1] != len(cs) - 1:
                raise KhgError("HmmTopology::Check(), pdf_classes are expected to be contiguous and start from zero.")

    # pickle: (phones, phone2idx, entries)  python/csrc/hmm-topology.cc:83-93
    def __getstate__(self):
        return (self._phones, self._phone2idx, self._entries)

    def __setstate__(self, t):
        self._phones, self._phone2idx, self._entries = list(t[0]), list(t[1]), [list(e) for e in t[2]]
