"""Accumulators and the M-step -- the C++ classes of csrc/khg_host_gmm.{hpp,cpp} (mirrors of csrc/model-common.{h,cc},
csrc/mle-diag-gmm.{h,cc}, csrc/mle-am-diag-gmm.{h,cc}) under the names of python/csrc/model-common.cc, mle-diag-gmm.cc,
mle-am-diag-gmm.cc.  Statistics are fp64 (the reference's DoubleVector / DoubleMatrix); accumulation from features runs on
the GPU (K3), the M-step in host C++ (khg_mle_am_diag_gmm_update) or on the device (K4).  What is left here: the
GmmUpdateFlags enum type and the glue of the device M-step."""
import enum
from typing import List

import numpy as np

from . import device  # noqa: F401
from . import _kaldi_hmm_gmm_amd as _ext
from ._lib import KhgError
from .diag_gmm import AmDiagGmm

AccumAmDiagGmm = _ext.AccumAmDiagGmm
AccumDiagGmm = _ext.AccumDiagGmm
MleDiagGmmOptions = _ext.MleDiagGmmOptions
MapDiagGmmOptions = _ext.MapDiagGmmOptions
ml_objective = _ext.ml_objective


class GmmUpdateFlags(enum.IntFlag):   # csrc/model-common.h:18-26
    kGmmMeans = 0x001
    kGmmVariances = 0x002
    kGmmWeights = 0x004
    kGmmTransitions = 0x008
    kGmmAll = 0x00F


kGmmMeans, kGmmVariances, kGmmWeights, kGmmTransitions, kGmmAll = (GmmUpdateFlags.kGmmMeans, GmmUpdateFlags.kGmmVariances,
                                                                     GmmUpdateFlags.kGmmWeights, GmmUpdateFlags.kGmmTransitions,
                                                                     GmmUpdateFlags.kGmmAll)


def str_to_gmm_flags(s: str) -> GmmUpdateFlags:   # csrc/model-common.cc:99-124
    return GmmUpdateFlags(_ext.str_to_gmm_flags(s))


def gmm_flags_to_str(flags) -> str:   # :126-145
    return _ext.gmm_flags_to_str(int(flags))


def augment_gmm_flags(flags) -> int:   # :72-85
    return _ext.augment_gmm_flags(int(flags))


def get_split_targets(state_occs, target_components: int, power: float, min_count: float) -> List[int]:
    """csrc/model-common.cc:29-70: max-heap on occ^power / #components with a min-count stop."""
    return _ext.get_split_targets(np.asarray(state_occs, np.float32), int(target_components), power, min_count)


def _flat_update(opts, gauss_off, occ, mean_acc, var_acc, acc_flags, flags, w, miv, iv):
    """khg_mle_am_diag_gmm_update over flat arrays -> (new_off, w, gc, miv, iv, objf_change, count, floored_elements,
    floored_gaussians, removed)."""
    return _ext.flat_update(opts, np.asarray(gauss_off, np.int32), occ, mean_acc, var_acc, int(acc_flags), int(flags), w, miv, iv)


def mle_diag_gmm_update(config, diag_gmm_acc, flags, gmm):
    """csrc/mle-diag-gmm.cc:243-390 -> (objf_change, count, floored_elements, floored_gaussians, removed)."""
    return _ext.mle_diag_gmm_update(config, diag_gmm_acc, int(flags), gmm)


def map_diag_gmm_update(config, diag_gmm_acc, flags, gmm):
    """csrc/mle-diag-gmm.cc:392-477 -> (objf_change, count)."""
    return _ext.map_diag_gmm_update(config, diag_gmm_acc, int(flags), gmm)


def map_am_diag_gmm_update(config, amdiag_gmm_acc, flags, am_gmm):
    """csrc/mle-am-diag-gmm.cc:204-227 -> (objf_change, count)."""
    return _ext.map_am_diag_gmm_update(config, amdiag_gmm_acc, int(flags), am_gmm)


def mle_am_diag_gmm_update(config, amdiag_gmm_acc, flags, am_gmm):
    """csrc/mle-am-diag-gmm.cc:153-202 -> (objf_change, count); am_gmm is updated in place."""
    return _ext.mle_am_diag_gmm_update(config, amdiag_gmm_acc, int(flags), am_gmm)


def mle_am_diag_gmm_update_device(config, device_accs, flags, device_model, am_gmm: AmDiagGmm = None):
    """mle_am_diag_gmm_update (csrc/mle-am-diag-gmm.cc:153-202) run on the GPU from the DeviceAccs block where
    K3 / the all-reduce left the statistics (khg_model_mle_update, K4): `device_model` is updated in place and
    is ready for the next loglikes / align / acc_stats pass without any accumulator download or parameter
    upload.  am_gmm (optional) receives the new parameters (one 2*sumG*dim float download) -- pass it on the
    iterations that write or mix up the model, leave it out otherwise.  -> (objf_change, count)."""
    r = device_model.mle_update(device_accs, config, int(flags) & 0x7)
    if r["removed"]:
        device_accs.relayout(device_model)
    if am_gmm is not None:
        d = device_model.download()
        if am_gmm.num_pdfs != device_model.num_pdfs:
            raise KhgError("am_gmm->NumPdfs() does not match the device model")
        am_gmm.set_flat(d["gauss_off"], d["weights"], d["gconsts"], d["means_invvars"], d["inv_vars"])
    return r["objf_change"], r["count"]
