"""HmmState / HmmTopology -- the C++ classes of csrc/khg_host_hmm.{hpp,cpp} (mirrors of csrc/hmm-topology.{h,cc}) under the
names of python/csrc/hmm-topology.cc: Kaldi text format, Check(), pickling as (phones, phone2idx, entries).  What is added
here is the stream I/O over kaldi_io (text and Kaldi binary, csrc/hmm-topology.cc:23-282), attached to the C++ class."""
import numpy as np

from . import device  # noqa: F401
from ._kaldi_hmm_gmm_amd import HmmState, HmmTopology, kNoPdf  # noqa: F401
from ._lib import KhgError


def _write(self, w) -> None:
    if not w.binary:
        w.raw(str(self))
        return
    hmm = self.is_hmm
    phones, phone2idx, entries = self._phones, self._phone2idx, self._entries
    w.token("<Topology>")
    w.int_vector(phones)
    w.int_vector(phone2idx)
    if not hmm:
        w.int32(-1)                 # marks the extended format with SelfLoopPdfClass
    w.int32(len(entries))
    for entry in entries:
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
    r.expect("<Topology>")
    phones = r.int_vector()
    phone2idx = r.int_vector()
    n = r.int32()
    hmm = True
    if n == -1:
        hmm = False
        n = r.int32()
    entries = []
    for _ in range(n):
        entry = []
        for _ in range(r.int32()):
            fwd = r.int32()
            st = HmmState(fwd, fwd if hmm else r.int32())
            tr = []
            for _ in range(r.int32()):
                dst = r.int32()
                tr.append((dst, float(np.float32(r.float32()))))
            st.transitions = tr
            entry.append(st)
        entries.append(entry)
    r.expect("</Topology>")
    self._set_state(list(phones), list(phone2idx), entries)
    self.check()


def write(self, binary: bool, filename: str) -> None:
    from . import kaldi_io
    w = kaldi_io.Writer(binary)
    self._write(w)
    kaldi_io.write_file(filename, binary, w.getvalue())


def read_file(self, filename: str) -> None:
    from . import kaldi_io
    self._read(kaldi_io.read_file(filename))


HmmTopology._write, HmmTopology._read, HmmTopology.write, HmmTopology.read_file = _write, _read, write, read_file
