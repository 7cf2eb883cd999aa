"""ctypes binding of libkhg_hip.so (C-ABI: include/khg_hip.h).

The library is the product: there is no Python/CPU fallback.  If it is missing the import
fails loudly; if no gfx950 GPU is present every compute call raises (khg_ctx_create fails).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KHG_LIBRARY") or os.path.join(_HERE, "libkhg_hip.so")   # KHG_LIBRARY: A/B builds (ctypes binding only)


class KhgError(RuntimeError):
    """Mirror of the reference's std::runtime_error -> Python RuntimeError (csrc/log.h:46-53)."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). kaldi_hmm_gmm_amd has no CPU fallback."
        )
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same
    # SONAME as /opt/rocm's, which this library is linked against).  Loaded torch-first, the dynamic linker resolves
    # our libamdhip64.so.7 to the copy already in the process and everything shares one runtime (streams, RCCL
    # tensors aliasing our buffers).  Loaded the other way round, torch later brings in its bundled copy as a SECOND
    # runtime, whose HSA layer cannot open the device any more ("No HIP GPUs are available").  So when torch is
    # installed, let it load its runtime first; without torch the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return C.CDLL(LIB_PATH)


lib = _load()

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class AlignConfigC(C.Structure):
    _fields_ = [
        ("beam", C.c_float),
        ("retry_beam", C.c_float),
        ("careful", C.c_int32),
        ("acoustic_scale", C.c_float),
        ("max_active", C.c_int32),
        ("min_active", C.c_int32),
        ("beam_delta", C.c_float),
        ("hash_ratio", C.c_float),
        ("like_scale", C.c_float),
    ]


class MleOptionsC(C.Structure):
    _fields_ = [
        ("min_gaussian_weight", C.c_float),
        ("min_gaussian_occupancy", C.c_float),
        ("min_variance", C.c_double),
        ("remove_low_count_gaussians", C.c_int32),
        ("variance_floor_vector", C.POINTER(C.c_double)),
    ]


# every symbol include/khg_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "khg_last_error": (C.c_char_p, []),
    "khg_version": (C.c_int, []),
    "khg_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "khg_ctx_destroy": (C.c_int, [vp]),
    "khg_ctx_sync": (C.c_int, [vp]),
    "khg_ctx_set_timing": (C.c_int, [vp, C.c_int]),
    "khg_ctx_set_k1_form": (C.c_int, [vp, C.c_int]),
    "khg_ctx_set_option": (C.c_int, [vp, C.c_int, C.c_int]),
    "khg_ctx_get_option": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "khg_ctx_get_timings": (C.c_int, [vp, C.c_char_p, C.c_int64, c_f32p, C.c_int32, c_i32p]),
    "khg_model_create": (C.c_int, [vp, C.c_int32, C.c_int32, c_i32p, c_f32p, c_f32p, c_f32p, C.POINTER(vp)]),
    "khg_model_destroy": (C.c_int, [vp]),
    "khg_tm_create": (C.c_int, [vp, C.c_int32, c_i32p, C.POINTER(vp)]),
    "khg_tm_set_trans_cost": (C.c_int, [vp, c_f32p]),
    "khg_tm_destroy": (C.c_int, [vp]),
    "khg_utts_create": (
        C.c_int,
        [vp, vp, C.c_int32, C.c_int32, c_i64p, c_f32p, vp, c_i64p, c_i32p, c_i64p, c_i32p, c_i32p, c_f32p, c_i32p,
         c_f32p, C.POINTER(vp)],
    ),
    "khg_utts_destroy": (C.c_int, [vp]),
    "khg_utts_num_pdfs": (C.c_int, [vp, c_i64p]),
    "khg_utts_pdfs": (C.c_int, [vp, c_i32p]),
    "khg_utts_pdf_first": (C.c_int, [vp, c_i32p]),
    "khg_loglikes": (C.c_int, [vp, vp, vp]),
    "khg_loglikes_reachable": (C.c_int, [vp, vp, vp]),
    "khg_loglikes_band": (C.c_int, [vp, vp, vp]),
    "khg_utts_pdf_last": (C.c_int, [vp, c_i32p]),
    "khg_loglikes_layout": (C.c_int, [vp, c_i64p, c_i64p]),
    "khg_loglikes_download": (C.c_int, [vp, vp, c_f32p]),
    "khg_loglikes_upload": (C.c_int, [vp, vp, c_f32p]),
    "khg_utts_set_pdf_list": (C.c_int, [vp, C.c_int32, c_i32p]),
    "khg_utts_features_changed": (C.c_int, [vp]),
    "khg_comm_info": (C.c_int, [vp, c_i32p, c_i32p, c_i32p]),
    "khg_align_config_default": (None, [C.POINTER(AlignConfigC)]),
    "khg_align": (C.c_int, [vp, vp, vp, C.POINTER(AlignConfigC), c_i32p, c_i32p, c_i64p, C.c_int64, c_f32p, c_i32p]),
    "khg_ali_upload": (C.c_int, [vp, vp, c_i32p]),
    "khg_ali_download": (C.c_int, [vp, vp, c_i32p]),
    "khg_accs_create": (C.c_int, [vp, vp, vp, C.POINTER(vp)]),
    "khg_accs_destroy": (C.c_int, [vp]),
    "khg_accs_zero": (C.c_int, [vp, vp]),
    "khg_accs_size": (C.c_int, [vp, c_i64p]),
    "khg_accs_device_ptr": (C.c_int, [vp, C.POINTER(vp)]),
    "khg_accs_download": (C.c_int, [vp, vp, c_f64p]),
    "khg_accs_upload": (C.c_int, [vp, vp, c_f64p]),
    "khg_acc_stats": (C.c_int, [vp, vp, vp, vp, C.c_float, vp]),
    "khg_accs_allreduce": (C.c_int, [vp, vp, vp]),
    "khg_accs_allreduce_range": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, vp]),
    "khg_acc_stats_reduce": (C.c_int, [vp, vp, vp, vp, C.c_float, vp, vp, C.c_int32]),
    "khg_accs_allreduce_f32": (C.c_int, [vp, vp, vp]),
    "khg_comm_unique_id": (C.c_int, [vp]),
    "khg_comm_create": (C.c_int, [vp, C.c_int32, C.c_int32, vp, C.POINTER(vp)]),
    "khg_comm_destroy": (C.c_int, [vp]),
    "khg_compute_gconsts": (C.c_int, [C.c_int32, C.c_int32, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p]),
    "khg_mle_options_default": (None, [C.POINTER(MleOptionsC)]),
    "khg_mle_am_diag_gmm_update": (
        C.c_int,
        [C.POINTER(MleOptionsC), C.c_int32, C.c_int32, c_i32p, c_f64p, c_f64p, c_f64p, C.c_uint16, C.c_uint16, c_f32p,
         c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p],
    ),
    "khg_careful_graph": (C.c_int, [C.c_int32, C.c_int32, c_i64p, c_i32p, c_i32p, c_f32p, c_i32p, c_f32p, c_i32p, c_i32p, c_i64p, c_i32p,
                                    c_i32p, c_f32p, c_i32p, c_f32p]),
    "khg_diag_gmm_merge": (C.c_int, [c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p]),
    "khg_model_set_weights": (C.c_int, [vp, vp, c_f32p]),
    "khg_model_mle_update": (C.c_int, [vp, vp, vp, C.POINTER(MleOptionsC), C.c_uint16, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p]),
    "khg_model_mle_update_sharded": (C.c_int, [vp, vp, vp, C.POINTER(MleOptionsC), C.c_uint16, vp, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                              c_i32p, c_i32p, c_i32p]),
    "khg_model_mle_update_range": (C.c_int, [vp, vp, vp, C.POINTER(MleOptionsC), C.c_uint16, C.c_int32, C.c_int32]),
    "khg_model_mle_rows_download": (C.c_int, [vp, vp, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, vp]),
    "khg_model_mle_rows_upload": (C.c_int, [vp, vp, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, vp]),
    "khg_model_mle_update_finish": (C.c_int, [vp, vp, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p]),
    "khg_model_split": (C.c_int, [vp, vp, c_i32p, C.c_float, c_f32p, C.c_int64]),
    "khg_model_merge": (C.c_int, [vp, vp, c_i32p]),
    "khg_model_num_gauss": (C.c_int, [vp, C.POINTER(C.c_int64), c_i32p]),
    "khg_model_invalidate": (C.c_int, [vp]),
    "khg_model_download": (C.c_int, [vp, vp, c_f32p, c_f32p, c_f32p, c_f32p]),
    "khg_accs_relayout": (C.c_int, [vp, vp, vp]),
    "khg_accs_download_trans": (C.c_int, [vp, vp, c_f64p, c_f64p]),
    "khg_accs_download_range": (C.c_int, [vp, vp, C.c_int64, C.c_int64, c_f64p]),
    "khg_model_scale_weights": (C.c_int, [vp, vp, C.c_int32, c_i32p, C.c_float]),
    "khg_transition_mle_update": (
        C.c_int,
        [C.c_int32, c_i32p, c_i32p, c_f64p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p],
    ),
    "khg_scaled_trans_cost": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_i32p, c_u8p, C.c_float, C.c_float, c_f32p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc):
    if rc != 0:
        raise KhgError(lib.khg_last_error().decode("utf-8", "replace"))


def ptr(a, ctype):
    """numpy array -> typed pointer (None passes NULL); the caller keeps `a` alive."""
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def as_np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
