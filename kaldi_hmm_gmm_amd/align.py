"""Alignment API -- the C++ of csrc/khg_host_align.{hpp,cpp} and csrc/khg_host_fst.{hpp,cpp} (mirrors of
csrc/decoder-wrappers.{h,cc}, csrc/faster-decoder.{h,cc}, csrc/decodable-am-diag-gmm.h, csrc/decodable-itf.h,
csrc/hmm-utils.cc:465-493) under their pybind names (python/csrc/decoder-wrappers.cc, faster-decoder.cc,
decodable-am-diag-gmm.cc, decodable-itf.cc, hmm-utils.cc): AlignConfig, FasterDecoderOptions, DecodableInterface,
DecodableAmDiagGmmUnmapped / Scaled, add_transition_probs, align_utterance_wrapper, the batched align_batch, and FasterDecoder with
its linear best-path lattice.  The work is done by K1 (log-likes) + K2 (Viterbi) through the C-ABI.  This module re-exports them."""
from . import device  # noqa: F401
from ._kaldi_hmm_gmm_amd import (AlignConfig, DecodableAmDiagGmmScaled, DecodableAmDiagGmmUnmapped, DecodableInterface,  # noqa: F401
                                 FasterDecoder, FasterDecoderOptions, LatticeArc, LatticeWeight, LinearLattice,
                                 add_transition_probs, align_batch, align_utterance_wrapper)
from .device import ALIGN_ERROR, ALIGN_RETRIED, INT32_MAX  # noqa: F401
