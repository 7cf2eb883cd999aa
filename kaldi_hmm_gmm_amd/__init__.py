"""kaldi_hmm_gmm_amd -- MI355X-native HMM-GMM EM hot path (K1 log-likes on the fp16 matrix cores at fp32 accuracy, K2 Viterbi
forced alignment, K3 sufficient statistics (posteriors on fp16 / fp32 MFMA, sums on fp64 MFMA), K4 device M-step, C1 RCCL sum) behind the names of the reference's
pybind11 module `kaldi_hmm_gmm` (python/kaldi_hmm_gmm/__init__.py) and of its scripts/*.py.

Importing this package loads libkhg_hip.so; there is no CPU fallback."""
from . import _lib  # noqa: F401  (fails loudly when libkhg_hip.so is missing)
from ._lib import KhgError  # noqa: F401
from .align import (AlignConfig, DecodableAmDiagGmmScaled, DecodableAmDiagGmmUnmapped, DecodableInterface, FasterDecoder,  # noqa: F401
                    FasterDecoderOptions, LatticeArc, LatticeWeight, LinearLattice, add_transition_probs, align_batch,
                    align_utterance_wrapper)
from .context_dep import (ContextDependency, ContextDependencyInterface, monophone_context_dependency,  # noqa: F401
                          monophone_context_dependency_shared)
from .device import (ALIGN_DONE, ALIGN_ERROR, ALIGN_EXACT_DP, ALIGN_FALLBACK, ALIGN_RETRIED, Comm, Context, DeviceAccs,  # noqa: F401
                     DeviceModel, DeviceTransitions, UtteranceSet)
from .diag_gmm import AmDiagGmm, DiagGmm  # noqa: F401
from .fst import StdArc, StdVectorFst, modify_graph_for_careful_alignment  # noqa: F401
from .hmm_topology import HmmState, HmmTopology  # noqa: F401
from .mle import (kGmmAll, kGmmMeans, kGmmTransitions, kGmmVariances, kGmmWeights)  # noqa: F401  (py::enum_::export_values, model-common.cc:12-20)
from .mle import (AccumAmDiagGmm, AccumDiagGmm, GmmUpdateFlags, MapDiagGmmOptions, MleDiagGmmOptions, augment_gmm_flags,  # noqa: F401
                  get_split_targets, gmm_flags_to_str, map_am_diag_gmm_update, map_diag_gmm_update, ml_objective, mle_am_diag_gmm_update,
                  mle_am_diag_gmm_update_device, mle_diag_gmm_update,
                  str_to_gmm_flags)
from .resident import ResidentEm  # noqa: F401
from .scripts import (gmm_acc_stats_ali, gmm_acc_stats_ali_batch, gmm_align_compiled, gmm_align_compiled_batch,  # noqa: F401
                      gmm_boost_silence, gmm_est, gmm_info, gmm_init_mono)
from .training_graph import (TrainingGraphCompiler, TrainingGraphCompilerOptions, equal_align, generate_hmm_topo,  # noqa: F401
                             make_lexicon_fst_with_silence)
from .transition_model import (MleTransitionUpdateConfig, TransitionInformation, TransitionModel, TransitionModelTuple,  # noqa: F401
                               get_pdfs_for_phones)
