"""Device-resident objects of the batched EM path: the C++ classes of `_kaldi_hmm_gmm_amd` (csrc/khg_pybind.cpp -- the
counterpart of the reference's `_kaldi_hmm_gmm`, python/csrc/kaldi-hmm-gmm.cc:35-69), which own the C-ABI handles of
libkhg_hip.so (include/khg_hip.h).  Batched counterparts of what scripts/gmm_align_compiled.py and
scripts/gmm_acc_stats_ali.py (reference) do one utterance / one frame at a time.

There is ONE binding: this module only re-exports the extension's classes (round 2 kept ctypes twins beside them).  The raw
C-ABI stays reachable through `_lib` (ctypes prototypes of every exported symbol) for the tests that exercise it directly.
"""
from . import _lib
from . import _kaldi_hmm_gmm_amd as _ext        # built by csrc/Makefile; no fallback: without it the package does not import

INT32_MAX = 2**31 - 1

ALIGN_DONE = 0
ALIGN_ERROR = 1
ALIGN_RETRIED = 2
ALIGN_EXACT_DP = 4
ALIGN_FALLBACK = 8

_ext._set_error_class(_lib.KhgError)
BINDING = "pybind11"

Context = _ext.Context                      # khg_ctx: one per process / GPU (+ set_k1_form / set_option / get_option, timings)
Comm = _ext.Comm                            # an RCCL communicator owned by the library (khg_comm_create)
DeviceModel = _ext.DeviceModel              # AmDiagGmm resident in HBM (+ mle_update, split, scale_weights)
DeviceTransitions = _ext.DeviceTransitions  # TransitionIdToPdf + scaled transition costs
UtteranceSet = _ext.UtteranceSet            # features + compiled graphs of a shard; loglikes / align / acc_stats


class DeviceAccs(_ext.DeviceAccs):
    """AccumAmDiagGmm + transition stats as one fp64 device buffer (C++ class; as_torch is the only Python part)."""

    def as_torch(self):
        """Zero-copy torch view of the device buffer (for torch.distributed.all_reduce over RCCL)."""
        import torch

        class _W:
            pass

        w = _W()
        w.__cuda_array_interface__ = {"shape": (self.size,), "typestr": "<f8", "data": (self.device_ptr(), False),
                                      "version": 2, "strides": None}
        t = torch.as_tensor(w, device=f"cuda:{self.ctx.device}")
        t._khg_keepalive = self
        return t
