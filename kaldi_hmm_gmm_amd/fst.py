"""StdVectorFst / StdArc (tropical semiring) with the kaldifst method names the reference's scripts use
(scripts/gmm_align_compiled.py:6-41, egs/yesno/train.py:70-108): the `kaldifst` package that registers
fst::VectorFst<StdArc> for the reference is not available offline, so the decoding-graph argument of this API is this class --
C++ in the extension (csrc/khg_host_fst.{hpp,cpp}), like the graph helpers next to it.  Final weight +inf ==
TropicalWeight::Zero() (not final).  `arcs(state)` returns copies (the container owns its arcs)."""
from . import device  # noqa: F401
from ._kaldi_hmm_gmm_amd import (StdArc, StdVectorFst, concat_graphs, kNoStateId,  # noqa: F401
                                 modify_graph_for_careful_alignment)
