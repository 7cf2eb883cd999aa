"""Lazily created default device context + small helpers that run single-object API calls
(DiagGmm.log_likelihood, AccumAmDiagGmm.accumulate_for_gmm, ...) through the same HIP kernels
the batched path uses.  There is no CPU implementation behind these."""
import os

import numpy as np

from . import _kaldi_hmm_gmm_amd as _ext
from .device import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet

_ctx = None


def default_context() -> Context:
    global _ctx
    if _ctx is None:
        set_default_context(Context(int(os.environ.get("KHG_DEVICE", "0"))))
    return _ctx


def set_default_context(ctx: Context):
    """The context the single-object calls use (DiagGmm.log_likelihood, AccumDiagGmm.accumulate_from_diag, align_batch ...):
    shared with the C++ host classes (khg::SetDefaultCtx)."""
    global _ctx
    _ctx = ctx
    _ext.set_default_context(ctx)


def loglikes(gauss_off, gconsts, means_invvars, inv_vars, feats, pdfs):
    """K1: -> float32 [len(pdfs), N] log-likelihoods of every frame under the listed pdfs."""
    ctx = default_context()
    feats = np.ascontiguousarray(feats, np.float32)
    if feats.ndim == 1:
        feats = feats[None, :]
    dm = DeviceModel(ctx, gauss_off, gconsts, means_invvars, inv_vars)
    us = UtteranceSet(ctx, None, np.array([0, feats.shape[0]], np.int64), feats)
    us.set_pdf_list(np.asarray(pdfs, np.int32))
    us.loglikes(dm)
    out = us.download_loglikes()[0]
    us.close(); dm.close()
    return out


def acc_stats(gauss_off, gconsts, means_invvars, inv_vars, feats, frame_pdf, weight=1.0):
    """K3 on explicit per-frame pdf ids (transition-id = pdf+1): -> dict of fp64 statistics."""
    ctx = default_context()
    feats = np.ascontiguousarray(feats, np.float32)
    if feats.ndim == 1:
        feats = feats[None, :]
    P = len(gauss_off) - 1
    dm = DeviceModel(ctx, gauss_off, gconsts, means_invvars, inv_vars)
    tm = DeviceTransitions(ctx, np.concatenate([[0], np.arange(P)]).astype(np.int32))
    us = UtteranceSet(ctx, None, np.array([0, feats.shape[0]], np.int64), feats)
    us.upload_ali(np.asarray(frame_pdf, np.int32) + 1)
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs, weight)
    out = accs.download()
    accs.close(); us.close(); tm.close(); dm.close()
    return out
