"""DiagGmm / AmDiagGmm -- the C++ classes of csrc/khg_host_gmm.{hpp,cpp} (mirrors of the reference's csrc/diag-gmm.{h,cc} and
csrc/am-diag-gmm.{h,cc}) bound in csrc/khg_py_host.cpp with the names and signatures of python/csrc/diag-gmm.cc /
am-diag-gmm.cc.  Parameters are fp32 in the reference's exponential form; every likelihood / posterior evaluation runs on the
GPU through the C-ABI.  This module only re-exports them."""
from . import device  # noqa: F401  (registers KhgError with the extension)
from ._kaldi_hmm_gmm_amd import AmDiagGmm, DiagGmm  # noqa: F401
