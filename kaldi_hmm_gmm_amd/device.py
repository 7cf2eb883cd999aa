"""Device-resident objects of the batched EM path: thin owners of the C-ABI handles.

Batched counterparts of what scripts/gmm_align_compiled.py and scripts/gmm_acc_stats_ali.py
(reference) do one utterance / one frame at a time.
"""
import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib, ptr

INT32_MAX = 2**31 - 1

ALIGN_DONE = 0
ALIGN_ERROR = 1
ALIGN_RETRIED = 2
ALIGN_EXACT_DP = 4
ALIGN_FALLBACK = 8


class Context:
    """khg_ctx: one per process / GPU.  `stream` may be a raw hipStream_t handle (int), e.g.
    torch.cuda.current_stream().cuda_stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.h = C.c_void_p()
        check(lib.khg_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self.h)))
        self.device = device

    def sync(self):
        check(lib.khg_ctx_sync(self.h))

    K1_FORMS = {"auto": 0, "bf16x3": 1, "fp32": 2, "pdf": 2, "utt": 3, "f16x2": 4, "f16x2s": 5}

    def set_k1_form(self, form: str):
        """Arithmetic / tiling of the log-likelihood kernel (khg_ctx_set_k1_form): "auto" (= "bf16x3"), "bf16x3", "pdf"
        (fp32 MFMA, pdf-major), "utt" (fp32 MFMA, utterance-major)."""
        check(lib.khg_ctx_set_k1_form(self.h, Context.K1_FORMS[form]))

    def set_timing(self, on: bool):
        check(lib.khg_ctx_set_timing(self.h, int(on)))

    def timings(self):
        """-> list of (kernel name, ms) recorded since the last call (HIP events on the ctx stream)."""
        cap = 4096
        names = C.create_string_buffer(1 << 16)
        ms = np.zeros(cap, np.float32)
        n = C.c_int32()
        check(lib.khg_ctx_get_timings(self.h, names, len(names), ptr(ms, C.c_float), cap, C.byref(n)))
        nm = names.value.decode().split("\n")[: n.value]
        return list(zip(nm, ms[: n.value].tolist()))

    def close(self):
        if self.h:
            lib.khg_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """An RCCL communicator owned by the library (khg_comm_create): rank 0 makes the 128-byte id
    (Comm.unique_id()), the caller ships it to every rank, every rank constructs Comm(ctx, nranks, rank, id)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(lib.khg_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx: "Context", nranks: int, rank: int, uid: bytes):
        assert len(uid) == Comm.ID_BYTES
        self.nranks, self.rank = int(nranks), int(rank)
        self.h = C.c_void_p()
        check(lib.khg_comm_create(ctx.h, self.nranks, self.rank, C.c_char_p(uid), C.byref(self.h)))

    def close(self):
        if self.h:
            lib.khg_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceModel:
    """AmDiagGmm uploaded as the K1 tile image + K3 row-major copy."""

    def __init__(self, ctx: Context, gauss_off, gconsts, means_invvars, inv_vars, weights=None):
        self.ctx = ctx
        go = _lib.as_np(gauss_off, np.int32)
        gc = _lib.as_np(gconsts, np.float32)
        miv = _lib.as_np(means_invvars, np.float32)
        iv = _lib.as_np(inv_vars, np.float32)
        self.num_pdfs = go.shape[0] - 1
        self.dim = miv.shape[1]
        self.gauss_off = go
        assert miv.shape == iv.shape and miv.shape[0] == go[-1] == gc.shape[0]
        self.h = C.c_void_p()
        check(lib.khg_model_create(ctx.h, self.num_pdfs, self.dim, ptr(go, C.c_int32), ptr(gc, C.c_float),
                                   ptr(miv, C.c_float), ptr(iv, C.c_float), C.byref(self.h)))
        if weights is not None:
            self.set_weights(weights)

    # -- K4: device M-step (khg_model_mle_update) ----------------------------------------------
    def set_weights(self, weights):
        """Mixture weights, only needed by mle_update (K1-K3 read them through gconsts)."""
        w = _lib.as_np(weights, np.float32)
        assert w.shape[0] == self.gauss_off[-1]
        check(lib.khg_model_set_weights(self.ctx.h, self.h, ptr(w, C.c_float)))

    def mle_update(self, accs: "DeviceAccs", opts=None, flags=0x7):
        """MleAmDiagGmmUpdate (csrc/mle-am-diag-gmm.cc:153-202) on the device, from the accumulators where K3
        (and the all-reduce) left them.  The handle is updated in place, gauss_off included.
        -> dict(objf_change, count, floored_elements, floored_gaussians, removed)."""
        o = _lib.MleOptionsC()
        if opts is None:
            lib.khg_mle_options_default(C.byref(o))
        else:
            o = opts._c() if hasattr(opts, "_c") else opts
        oc, cnt = C.c_float(), C.c_float()
        fe, fg, rm = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.khg_model_mle_update(self.ctx.h, self.h, accs.h, C.byref(o), C.c_uint16(int(flags) & 0xFFFF), C.byref(oc),
                                       C.byref(cnt), C.byref(fe), C.byref(fg), C.byref(rm)))
        if rm.value:
            go = np.zeros(self.num_pdfs + 1, np.int32)
            check(lib.khg_model_num_gauss(self.h, None, ptr(go, C.c_int32)))
            self.gauss_off = go
        return {"objf_change": oc.value, "count": cnt.value, "floored_elements": fe.value, "floored_gaussians": fg.value,
                "removed": rm.value}

    def split(self, targets, perturb_factor: float, randn):
        """Mixing up on the handle (khg_model_split = DiagGmm::Split per pdf, csrc/diag-gmm.cc:780-851): targets[p] components
        per pdf; randn: float32 [sum of new components, dim] injected normal deviates, consumed in (pdf, split) order."""
        t = _lib.as_np(targets, np.int32)
        r = _lib.as_np(randn, np.float32) if randn is not None else None
        check(lib.khg_model_split(self.ctx.h, self.h, ptr(t, C.c_int32), float(perturb_factor), ptr(r, C.c_float)))
        go = np.zeros(self.num_pdfs + 1, np.int32)
        check(lib.khg_model_num_gauss(self.h, None, ptr(go, C.c_int32)))
        self.gauss_off = go

    def scale_weights(self, pdfs, scale: float):
        """gmm_boost_silence (scripts/gmm_boost_silence.py:10-45) on the handle: weights of `pdfs` *= scale,
        their gconsts recomputed."""
        p = _lib.as_np(pdfs, np.int32)
        check(lib.khg_model_scale_weights(self.ctx.h, self.h, p.shape[0], ptr(p, C.c_int32), float(scale)))

    def download(self, weights=True):
        """-> dict(gauss_off, weights, gconsts, means_invvars, inv_vars) of the handle's current parameters."""
        G, D = int(self.gauss_off[-1]), self.dim
        w = np.zeros(G, np.float32) if weights else None
        gc = np.zeros(G, np.float32)
        miv = np.zeros((G, D), np.float32)
        iv = np.zeros((G, D), np.float32)
        check(lib.khg_model_download(self.ctx.h, self.h, ptr(w, C.c_float) if weights else None, ptr(gc, C.c_float),
                                     ptr(miv, C.c_float), ptr(iv, C.c_float)))
        return {"gauss_off": self.gauss_off.copy(), "weights": w, "gconsts": gc, "means_invvars": miv, "inv_vars": iv}

    def close(self):
        if self.h:
            lib.khg_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceTransitions:
    """TransitionIdToPdf table + the per-tid cost AddTransitionProbs adds to graph arcs."""

    def __init__(self, ctx: Context, id2pdf):
        self.ctx = ctx
        self.id2pdf = _lib.as_np(id2pdf, np.int32)
        self.num_tids = self.id2pdf.shape[0] - 1
        self.h = C.c_void_p()
        check(lib.khg_tm_create(ctx.h, self.num_tids, ptr(self.id2pdf, C.c_int32), C.byref(self.h)))

    def set_trans_cost(self, cost):
        if cost is None:
            check(lib.khg_tm_set_trans_cost(self.h, None))
        else:
            c = _lib.as_np(cost, np.float32)
            assert c.shape[0] == self.num_tids + 1
            check(lib.khg_tm_set_trans_cost(self.h, ptr(c, C.c_float)))

    def close(self):
        if self.h:
            lib.khg_tm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class UtteranceSet:
    """Features (+ decoding graphs) of a shard of utterances, resident in HBM.

    graphs: dict with state_off[U+1], start[U], arc_off[sumS+1], ilabel, olabel, weight,
    nextstate, final[sumS]  (CSR by source state, utterance-local nextstate), or None.
    feats: float32 [N, D] numpy array, or (device_ptr:int, keepalive) for data already in HBM.
    """

    def __init__(self, ctx: Context, tm: Optional[DeviceTransitions], frame_off, feats, dim=None, graphs=None):
        self.ctx = ctx
        self.frame_off = _lib.as_np(frame_off, np.int64)
        self.n_utt = self.frame_off.shape[0] - 1
        self._keep = []
        feats_h = None
        feats_d = None
        if isinstance(feats, tuple):
            feats_d, keep = feats
            self._keep.append(keep)
            assert dim is not None
        else:
            feats_h = _lib.as_np(feats, np.float32)
            dim = feats_h.shape[1]
            assert feats_h.shape[0] == self.frame_off[-1]
        self.dim = int(dim)
        g = graphs
        if g is not None:
            self._g = {k: _lib.as_np(g[k], dt) for k, dt in (
                ("state_off", np.int64), ("start", np.int32), ("arc_off", np.int64), ("ilabel", np.int32),
                ("olabel", np.int32), ("weight", np.float32), ("nextstate", np.int32), ("final", np.float32))}
            gg = self._g
            args = (ptr(gg["state_off"], C.c_int64), ptr(gg["start"], C.c_int32), ptr(gg["arc_off"], C.c_int64),
                    ptr(gg["ilabel"], C.c_int32), ptr(gg["olabel"], C.c_int32), ptr(gg["weight"], C.c_float),
                    ptr(gg["nextstate"], C.c_int32), ptr(gg["final"], C.c_float))
        else:
            args = (None,) * 8
        self.h = C.c_void_p()
        check(lib.khg_utts_create(ctx.h, tm.h if tm is not None else None, self.n_utt, self.dim,
                                  ptr(self.frame_off, C.c_int64), ptr(feats_h, C.c_float) if feats_h is not None else None,
                                  C.c_void_p(feats_d) if feats_d else None, *args, C.byref(self.h)))
        self._g = None  # the library keeps its own device copy

    # -- pdf lists / log-likes -------------------------------------------------------------
    def set_pdf_list(self, pdfs):
        p = _lib.as_np(pdfs, np.int32)
        check(lib.khg_utts_set_pdf_list(self.h, p.shape[0], ptr(p, C.c_int32)))

    def pdf_lists(self):
        off = np.zeros(self.n_utt + 1, np.int64)
        check(lib.khg_utts_num_pdfs(self.h, ptr(off, C.c_int64)))
        pdfs = np.zeros(max(int(off[-1]), 1), np.int32)
        check(lib.khg_utts_pdfs(self.h, ptr(pdfs, C.c_int32)))
        return off, pdfs[: int(off[-1])]

    def pdf_first_frames(self):
        """Per listed pdf: first frame a decoder token can read it at (khg_utts_pdf_first)."""
        off, _ = self.pdf_lists()
        first = np.zeros(max(int(off[-1]), 1), np.int32)
        check(lib.khg_utts_pdf_first(self.h, ptr(first, C.c_int32)))
        return first[: int(off[-1])]

    def loglikes(self, model: DeviceModel, reachable_only: bool = False):
        """K1.  reachable_only: skip the (pdf, frame) cells no decoder token can read (khg_loglikes_reachable);
        those cells of the score buffer are then unspecified."""
        check((lib.khg_loglikes_reachable if reachable_only else lib.khg_loglikes)(self.ctx.h, model.h, self.h))

    def loglikes_layout(self):
        off = np.zeros(self.n_utt + 1, np.int64)
        tot = C.c_int64()
        check(lib.khg_loglikes_layout(self.h, ptr(off, C.c_int64), C.byref(tot)))
        return off, tot.value

    def download_loglikes(self):
        """-> list of [npdf_u, T_u] float32 arrays (padding columns stripped)."""
        off, tot = self.loglikes_layout()
        buf = np.zeros(max(tot, 1), np.float32)
        check(lib.khg_loglikes_download(self.ctx.h, self.h, ptr(buf, C.c_float)))
        poff, _ = self.pdf_lists()
        out = []
        for u in range(self.n_utt):
            T = int(self.frame_off[u + 1] - self.frame_off[u])
            tpad = (T + 31) // 32 * 32
            n = int(poff[u + 1] - poff[u])
            out.append(buf[off[u]: off[u] + n * tpad].reshape(n, tpad)[:, :T].copy())
        return out

    def upload_loglikes(self, mats):
        off, tot = self.loglikes_layout()
        buf = np.zeros(max(tot, 1), np.float32)
        for u, m in enumerate(mats):
            T = int(self.frame_off[u + 1] - self.frame_off[u])
            tpad = (T + 31) // 32 * 32
            n = m.shape[0]
            v = buf[off[u]: off[u] + n * tpad].reshape(n, tpad)
            v[:, :T] = m
        check(lib.khg_loglikes_upload(self.ctx.h, self.h, ptr(buf, C.c_float)))

    # -- alignment ---------------------------------------------------------------------------
    def align(self, tm: DeviceTransitions, beam=200.0, retry_beam=0.0, acoustic_scale=1.0, careful=False,
              max_active=INT32_MAX, min_active=20, beam_delta=0.5, hash_ratio=2.0, download=True):
        cfg = _lib.AlignConfigC(beam, retry_beam, int(careful), acoustic_scale, max_active, min_active, beam_delta,
                                hash_ratio)
        if not download:
            check(lib.khg_align(self.ctx.h, tm.h, self.h, C.byref(cfg), None, None, None, 0, None, None))
            return None
        if download == "summary":          # per-utterance likelihood + status only; alignments stay in HBM for K3
            like = np.zeros(self.n_utt, np.float32)
            status = np.zeros(self.n_utt, np.int32)
            check(lib.khg_align(self.ctx.h, tm.h, self.h, C.byref(cfg), None, None, None, 0, ptr(like, C.c_float),
                                ptr(status, C.c_int32)))
            return {"like": like, "status": status}
        N = int(self.frame_off[-1])
        ali = np.zeros(max(N, 1), np.int32)
        like = np.zeros(self.n_utt, np.float32)
        status = np.zeros(self.n_utt, np.int32)
        wcap = N + 16 * self.n_utt + 1024
        words = np.zeros(wcap, np.int32)
        woff = np.zeros(self.n_utt + 1, np.int64)
        check(lib.khg_align(self.ctx.h, tm.h, self.h, C.byref(cfg), ptr(ali, C.c_int32), ptr(words, C.c_int32),
                            ptr(woff, C.c_int64), wcap, ptr(like, C.c_float), ptr(status, C.c_int32)))
        return {"ali": ali[:N], "like": like, "status": status, "words": words[: int(woff[-1])], "words_off": woff}

    def upload_ali(self, ali):
        a = _lib.as_np(ali, np.int32)
        assert a.shape[0] == self.frame_off[-1]
        check(lib.khg_ali_upload(self.ctx.h, self.h, ptr(a, C.c_int32)))

    def download_ali(self):
        """The resident alignment (0 on the frames of utterances that failed to align)."""
        n = int(self.frame_off[-1])
        a = np.zeros(max(n, 1), np.int32)
        check(lib.khg_ali_download(self.ctx.h, self.h, ptr(a, C.c_int32)))
        return a[:n]

    def acc_stats(self, model: DeviceModel, tm: DeviceTransitions, accs: "DeviceAccs", weight: float = 1.0):
        check(lib.khg_acc_stats(self.ctx.h, model.h, tm.h, self.h, float(weight), accs.h))

    def close(self):
        if self.h:
            lib.khg_utts_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceAccs:
    """AccumAmDiagGmm + transition stats as one fp64 device buffer (see include/khg_hip.h)."""

    def __init__(self, ctx: Context, model: DeviceModel, tm: DeviceTransitions):
        self.ctx = ctx
        self.sumG = int(model.gauss_off[-1])
        self.dim = model.dim
        self.num_tids = tm.num_tids
        self.h = C.c_void_p()
        check(lib.khg_accs_create(ctx.h, model.h, tm.h, C.byref(self.h)))
        n = C.c_int64()
        check(lib.khg_accs_size(self.h, C.byref(n)))
        self.size = n.value

    def zero(self):
        check(lib.khg_accs_zero(self.ctx.h, self.h))

    def device_ptr(self) -> int:
        p = C.c_void_p()
        check(lib.khg_accs_device_ptr(self.h, C.byref(p)))
        return p.value

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

    def allreduce(self, comm: "Comm" = None, wire_fp32: bool = False):
        """C1 (khg_accs_allreduce): in-place sum of the block over the ranks of `comm`, enqueued on the context's
        stream behind K3.  wire_fp32: the fp32-wire tolerance experiment (khg_accs_allreduce_f32).  comm None =
        one-rank job."""
        fn = lib.khg_accs_allreduce_f32 if wire_fp32 else lib.khg_accs_allreduce
        check(fn(self.ctx.h, self.h, comm.h if comm is not None else None))

    def split(self, buf):
        G, D, nt = self.sumG, self.dim, self.num_tids
        o = 0
        occ = buf[o: o + G]; o += G
        mean = buf[o: o + G * D].reshape(G, D); o += G * D
        var = buf[o: o + G * D].reshape(G, D); o += G * D
        trans = buf[o: o + nt + 1]; o += nt + 1
        scal = buf[o: o + 8]
        return {"occ": occ, "mean_acc": mean, "var_acc": var, "trans_acc": trans, "total_frames": float(scal[0]),
                "total_log_like": float(scal[1])}

    def relayout(self, model: DeviceModel):
        """After DeviceModel.mle_update removed Gaussians: adopt the model's new gauss_off (and zero)."""
        check(lib.khg_accs_relayout(self.ctx.h, self.h, model.h))
        self.sumG = int(model.gauss_off[-1])
        n = C.c_int64()
        check(lib.khg_accs_size(self.h, C.byref(n)))
        self.size = n.value

    def download_range(self, first: int, count: int):
        out = np.zeros(max(count, 1), np.float64)
        check(lib.khg_accs_download_range(self.ctx.h, self.h, int(first), int(count), ptr(out, C.c_double)))
        return out[:count]

    def download_occ(self):
        """Per-Gaussian occupancies only (what the mix-up targets of scripts/gmm_est.py:66-70 need)."""
        return self.download_range(0, self.sumG)

    def download_trans(self):
        """Only the transition statistics and the scalar totals (what the host-side transition update needs)."""
        tr = np.zeros(self.num_tids + 1, np.float64)
        sc = np.zeros(8, np.float64)
        check(lib.khg_accs_download_trans(self.ctx.h, self.h, ptr(tr, C.c_double), ptr(sc, C.c_double)))
        return {"trans_acc": tr, "total_frames": float(sc[0]), "total_log_like": float(sc[1])}

    def download(self):
        buf = np.zeros(self.size, np.float64)
        check(lib.khg_accs_download(self.ctx.h, self.h, ptr(buf, C.c_double)))
        return self.split(buf)

    def upload(self, buf):
        b = _lib.as_np(buf, np.float64)
        assert b.shape[0] == self.size
        check(lib.khg_accs_upload(self.ctx.h, self.h, ptr(b, C.c_double)))

    def close(self):
        if self.h:
            lib.khg_accs_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the pybind11 host surface (csrc/khg_pybind.cpp) ----------------------------------------------------------------------
# The classes above are the ctypes twin of the C++ classes of `_kaldi_hmm_gmm_amd`, the module that corresponds to the
# reference's `_kaldi_hmm_gmm` (python/csrc/kaldi-hmm-gmm.cc:35-69).  When the extension is built (it is by
# `__graft_entry__.build()`), the package uses the C++ classes; KHG_BINDING=ctypes keeps the twins (same API, same C-ABI
# underneath -- either way every call ends in libkhg_hip.so, there is no other implementation).
import os as _os

BINDING = "ctypes"
if _os.environ.get("KHG_BINDING", "pybind11") != "ctypes":
    try:
        from . import _kaldi_hmm_gmm_amd as _ext
    except ImportError:
        _ext = None
    if _ext is not None:
        _ext._set_error_class(_lib.KhgError)
        _CtypesDeviceAccs = DeviceAccs
        Context, Comm, DeviceModel, DeviceTransitions, UtteranceSet = (_ext.Context, _ext.Comm, _ext.DeviceModel, _ext.DeviceTransitions,
                                                                   _ext.UtteranceSet)

        class DeviceAccs(_ext.DeviceAccs):
            """AccumAmDiagGmm + transition stats as one fp64 device buffer (C++ class; as_torch is the only Python part)."""
            as_torch = _CtypesDeviceAccs.as_torch

        BINDING = "pybind11"
