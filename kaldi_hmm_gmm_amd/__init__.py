"""MI355X-native HMM-GMM EM hot path (align + acc-stats + M-step) behind the reference's names."""
from . import _lib  # noqa: F401  (fails loudly when libkhg_hip.so is missing)
from .device import (ALIGN_DONE, ALIGN_ERROR, ALIGN_EXACT_DP, ALIGN_FALLBACK, ALIGN_RETRIED, Context, DeviceAccs,  # noqa: F401
                     DeviceModel, DeviceTransitions, UtteranceSet)
