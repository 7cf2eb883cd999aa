"""TransitionModel -- the C++ classes of csrc/khg_host_hmm.{hpp,cpp} (mirrors of csrc/transition-model.{h,cc} /
transition-information.h) under the names of python/csrc/transition-model.cc, transition-information.cc: the integer tables,
the M-step update (khg_transition_mle_update; MleUpdateShared for share_for_pdfs), the scaled transition costs, pickling as
the reference's 8-tuple.  What is added here is the stream I/O over kaldi_io (csrc/transition-model.cc:37-116, text and
binary), attached to the C++ class."""
import numpy as np

from . import device  # noqa: F401
from ._kaldi_hmm_gmm_amd import (MleTransitionUpdateConfig, TransitionInformation, TransitionModel,  # noqa: F401
                                 TransitionModelTuple, get_pdfs_for_phones)
from ._lib import KhgError
from .hmm_topology import HmmTopology


def _write(self, w) -> None:
    if not w.binary:
        w.raw(str(self))
        return
    hmm = self.topo.is_hmm
    w.token("<TransitionModel>")
    self.topo._write(w)
    w.token("<Triples>" if hmm else "<Tuples>")
    tuples = self.tuples
    w.int32(len(tuples))
    for t in tuples:
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
    topo = HmmTopology()
    topo._read(r)
    tok = r.token()
    if tok not in ("<Triples>", "<Tuples>"):
        raise KhgError(f"TransitionModel::Read, unexpected token {tok}")
    n = r.int32()
    tuples = []
    for _ in range(n):
        ph, hs, fp = r.int32(), r.int32(), r.int32()
        tuples.append(TransitionModelTuple(ph, hs, fp, r.int32() if tok == "<Tuples>" else fp))
    end = r.token()
    if end not in ("</Triples>", "</Tuples>"):
        raise KhgError(f"TransitionModel::Read, unexpected token {end}")
    r.expect("<LogProbs>")
    log_probs = np.asarray(r.float_vector(), np.float32).copy()
    r.expect("</LogProbs>")
    r.expect("</TransitionModel>")
    self._set_from_read(topo, tuples, log_probs)      # ComputeDerived, ComputeDerivedOfProbs, Check


def write(self, binary: bool, filename: str) -> None:
    from . import kaldi_io
    w = kaldi_io.Writer(binary)
    self._write(w)
    kaldi_io.write_file(filename, binary, w.getvalue())


def read(self, filename: str) -> None:
    from . import kaldi_io
    self._read(kaldi_io.read_file(filename))


TransitionModel._write, TransitionModel._read, TransitionModel.write, TransitionModel.read = _write, _read, write, read
