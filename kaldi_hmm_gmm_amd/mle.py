"""Accumulators and the M-step -- mirrors of csrc/model-common.{h,cc}, csrc/mle-diag-gmm.{h,cc},
csrc/mle-am-diag-gmm.{h,cc} (pybind: python/csrc/model-common.cc, mle-diag-gmm.cc,
mle-am-diag-gmm.cc).  Statistics are fp64 numpy arrays (the reference's DoubleVector/Matrix);
accumulation from features runs on the GPU (K3), the M-step in host C++ (khg_mle_am_diag_gmm_update)."""
import ctypes as C
import enum
import heapq
from typing import List

import numpy as np

from . import _gpu, _lib
from ._lib import KhgError, check, lib, ptr
from .diag_gmm import AmDiagGmm, DiagGmm

f32, f64 = np.float32, np.float64


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
    flags = 0
    table = {"m": 1, "v": 2, "w": 4, "t": 8, "a": 15}
    for ch in s:
        if ch not in table:
            raise KhgError(f"Invalid element '{ch}' of GmmFlagsType option string {s}")
        flags |= table[ch]
    return GmmUpdateFlags(flags)


def gmm_flags_to_str(flags) -> str:   # :126-145
    flags = int(flags)
    return "".join(c for c, b in (("m", 1), ("v", 2), ("w", 4), ("t", 8)) if flags & b)


def augment_gmm_flags(flags) -> int:   # :72-85
    flags = int(flags)
    if flags & ~0xF:
        raise KhgError("(flags & ~kGmmAll) == 0 assertion failed")
    if flags & 2:
        flags |= 1
    if flags & 1:
        flags |= 4
    if not flags & 4:
        flags |= 4
    return flags


def get_split_targets(state_occs, target_components: int, power: float, min_count: float) -> List[int]:
    """csrc/model-common.cc:29-70: max-heap on occ^power / #components with a min-count stop."""
    occs = np.asarray(state_occs, f32)
    P = occs.shape[0]
    # std::priority_queue<CountStats>: operator< compares occupancy / (num_components + 1e-10)
    heap = []
    order = 0
    for p in range(P):
        occ = float(f32(np.power(occs[p], f32(power), dtype=f32)))
        heapq.heappush(heap, (-(occ / (1 + 1.0e-10)), order, p, 1, occ)); order += 1
    num_gauss = P
    while num_gauss < target_components:
        key, _, p, nc, occ = heap[0]
        if occ == 0:
            break
        heapq.heappop(heap)
        if (nc + 1) * min_count >= float(occs[p]):
            occ = 0.0
        else:
            nc += 1
            num_gauss += 1
        heapq.heappush(heap, (-(occ / (nc + 1.0e-10)), order, p, nc, occ)); order += 1
    targets = [0] * P
    for _, _, p, nc, _ in heap:
        targets[p] = nc
    return targets


class MleDiagGmmOptions:   # csrc/mle-diag-gmm.h:23-45
    def __init__(self, min_gaussian_weight: float = 1.0e-05, min_gaussian_occupancy: float = 10.0,
                 min_variance: float = 0.001, remove_low_count_gaussians: bool = True, variance_floor_vector=None):
        self.variance_floor_vector = variance_floor_vector     # per-dimension floor, float64 [dim] (csrc/mle-diag-gmm.h:26-28)
        self.min_gaussian_weight = min_gaussian_weight
        self.min_gaussian_occupancy = min_gaussian_occupancy
        self.min_variance = min_variance
        self.remove_low_count_gaussians = remove_low_count_gaussians

    def __str__(self):
        return (f"MleDiagGmmOptions(min_gaussian_weight={self.min_gaussian_weight:g}, "
                f"min_gaussian_occupancy={self.min_gaussian_occupancy:g}, min_variance={self.min_variance:g}, "
                f"remove_low_count_gaussians={'True' if self.remove_low_count_gaussians else 'False'})")

    def _c(self):
        o = _lib.MleOptionsC(self.min_gaussian_weight, self.min_gaussian_occupancy, self.min_variance,
                             int(self.remove_low_count_gaussians), None)
        if self.variance_floor_vector is not None and len(self.variance_floor_vector):
            import ctypes as C
            self._vfv = np.ascontiguousarray(self.variance_floor_vector, np.float64)     # kept alive by the options object
            o.variance_floor_vector = self._vfv.ctypes.data_as(C.POINTER(C.c_double))
        return o


class AccumDiagGmm:
    """csrc/mle-diag-gmm.h:68-181: fp64 occupancy / mean / variance accumulators of one DiagGmm."""

    def __init__(self, gmm: DiagGmm = None, flags=0):
        self._flags = 0
        self.occupancy = np.zeros(0, f64)
        self.mean_accumulator = np.zeros((0, 0), f64)
        self.variance_accumulator = np.zeros((0, 0), f64)
        self._dim = 0
        if gmm is not None:
            self.resize(gmm.num_gauss, gmm.dim, flags)

    def resize(self, num_gauss, dim=None, flags=None):
        if isinstance(num_gauss, DiagGmm):   # resize(gmm, flags)
            num_gauss, dim, flags = num_gauss.num_gauss, num_gauss.dim, dim
        if not (num_gauss > 0 and dim > 0):
            raise KhgError("num_comp > 0 && dim > 0 assertion failed")
        self._flags = augment_gmm_flags(flags)
        self._dim = dim
        self.occupancy = np.zeros(num_gauss, f64)
        self.mean_accumulator = np.zeros((num_gauss, dim), f64) if self._flags & 1 else np.zeros((0, 0), f64)
        self.variance_accumulator = np.zeros((num_gauss, dim), f64) if self._flags & 2 else np.zeros((0, 0), f64)

    @property
    def num_gauss(self):
        return self.occupancy.shape[0]

    @property
    def dim(self):
        return self._dim

    @property
    def flags(self):
        return self._flags

    def _chk_flags(self, flags):
        if int(flags) & ~self._flags:
            raise KhgError("Flags in argument do not match the active accumulators")

    def set_zero(self, flags):
        self._chk_flags(flags)
        if int(flags) & 4:
            self.occupancy[:] = 0
        if int(flags) & 1:
            self.mean_accumulator[:] = 0
        if int(flags) & 2:
            self.variance_accumulator[:] = 0

    def scale(self, f: float, flags):
        self._chk_flags(flags)
        d = float(f32(f))
        if int(flags) & 4:
            self.occupancy *= d
        if int(flags) & 1:
            self.mean_accumulator *= d
        if int(flags) & 2:
            self.variance_accumulator *= d

    def accumulate_for_component(self, data, comp_index: int, weight: float):   # csrc/mle-diag-gmm.cc:100-121
        data = np.asarray(data, f32).reshape(-1)
        if self._flags & 1 and data.shape[0] != self._dim:
            raise KhgError("data.size() == Dim() assertion failed")
        if not comp_index < self.num_gauss:
            raise KhgError("comp_index < NumGauss() assertion failed")
        wt = float(f32(weight))
        self.occupancy[comp_index] += wt
        if self._flags & 1:
            self.mean_accumulator[comp_index] += data.astype(f64) * wt
            if self._flags & 2:
                self.variance_accumulator[comp_index] += ((data * data) * f32(wt)).astype(f64)

    def accumulate_from_posteriors(self, data, gauss_posteriors):   # :123-143
        data = np.asarray(data, f32).reshape(-1)
        post = np.asarray(gauss_posteriors, f32).reshape(-1)
        if self._flags & 1 and data.shape[0] != self._dim:
            raise KhgError("data.size() == Dim() assertion failed")
        if post.shape[0] != self.num_gauss:
            raise KhgError("posteriors.size() == NumGauss() assertion failed")
        self.occupancy += post.astype(f64)
        if self._flags & 1:
            self.mean_accumulator += np.outer(post, data).astype(f32).astype(f64)
            if self._flags & 2:
                self.variance_accumulator += np.outer(post, data * data).astype(f32).astype(f64)

    def accumulate_from_diag(self, gmm: DiagGmm, data, weight: float) -> float:   # :145-158, on the GPU (K3)
        if gmm.num_gauss != self.num_gauss or gmm.dim != self._dim:
            raise KhgError("gmm.NumGauss() == NumGauss() assertion failed")
        data = np.asarray(data, f32).reshape(-1)
        if data.shape[0] != self._dim:
            raise KhgError("data.size() == Dim() assertion failed")
        gmm._need_gconsts()
        st = _gpu.acc_stats(*gmm._as_model(False), data, [0], weight)
        self.occupancy += st["occ"]
        if self._flags & 1:
            self.mean_accumulator += st["mean_acc"]
        if self._flags & 2:
            self.variance_accumulator += st["var_acc"]
        w = float(f32(weight))
        return float(f32(st["total_log_like"] / w)) if w != 0 else 0.0

    def add_stats_for_component(self, g: int, occ: float, x_stats, x2_stats):
        if not g < self.num_gauss:
            raise KhgError("g < NumGauss() assertion failed")
        self.occupancy[g] += occ
        if self._flags & 1:
            self.mean_accumulator[g] += np.asarray(x_stats, f64)
        if self._flags & 2:
            self.variance_accumulator[g] += np.asarray(x2_stats, f64)

    def add(self, scale: float, acc: "AccumDiagGmm"):   # :176-188
        s = float(f32(scale))
        self.occupancy += acc.occupancy * s
        if self._flags & 1:
            self.mean_accumulator += acc.mean_accumulator * s
        if self._flags & 2:
            self.variance_accumulator += acc.variance_accumulator * s

    def smooth_stats(self, tau: float):   # csrc/mle-diag-gmm.cc:192-203 (tau "virtual counts" of the acc's own stats)
        tau = float(f32(tau))
        with np.errstate(divide="ignore", invalid="ignore"):
            sv = (self.occupancy + tau) / self.occupancy
        if self.mean_accumulator.size:
            self.mean_accumulator *= sv[:, None]
        if self.variance_accumulator.size:
            self.variance_accumulator *= sv[:, None]
        self.occupancy = self.occupancy + tau

    def smooth_with_accum(self, tau: float, src_acc: "AccumDiagGmm"):   # :209-226
        if src_acc.num_gauss != self.num_gauss or src_acc.dim != self._dim:
            raise KhgError("src_acc.NumGauss() == num_comp_ && src_acc.Dim() == dim_ assertion failed")
        tau = float(f32(tau))
        for i in range(self.num_gauss):
            so = src_acc.occupancy[i]
            if so != 0.0:    # can only smooth where the source saw data (the reference warns otherwise)
                self.occupancy[i] += tau
                self.mean_accumulator[i] += src_acc.mean_accumulator[i] * tau / so
                self.variance_accumulator[i] += src_acc.variance_accumulator[i] * tau / so

    def smooth_with_model(self, tau: float, gmm: DiagGmm):   # :228-241
        if gmm.num_gauss != self.num_gauss or gmm.dim != self._dim:
            raise KhgError("gmm.NumGauss() == num_comp_ && gmm.Dim() == dim_ assertion failed")
        tau = float(f32(tau))
        means = gmm.means.astype(f64)
        vars_ = gmm.vars.astype(f64)
        self.mean_accumulator += means * tau
        self.variance_accumulator += (vars_ + means * means) * tau
        self.occupancy = self.occupancy + tau

    def copy(self) -> "AccumDiagGmm":
        o = AccumDiagGmm()
        o._flags, o._dim = self._flags, self._dim
        o.occupancy = self.occupancy.copy()
        o.mean_accumulator = self.mean_accumulator.copy()
        o.variance_accumulator = self.variance_accumulator.copy()
        return o


def _flat_update(opts, gauss_off, occ, mean_acc, var_acc, acc_flags, flags, w, miv, iv):
    P = len(gauss_off) - 1
    D = miv.shape[1]
    go = _lib.as_np(gauss_off, np.int32)
    w = np.array(w, f32, copy=True); miv = np.array(miv, f32, copy=True); iv = np.array(iv, f32, copy=True)
    gc = np.zeros_like(w)
    new_off = np.zeros(P + 1, np.int32)
    oc, cnt = C.c_float(), C.c_float()
    fe, fg, rm = C.c_int32(), C.c_int32(), C.c_int32()
    o = opts._c()
    occ = _lib.as_np(occ, f64)
    ma = _lib.as_np(mean_acc, f64) if mean_acc is not None and mean_acc.size else None
    va = _lib.as_np(var_acc, f64) if var_acc is not None and var_acc.size else None
    check(lib.khg_mle_am_diag_gmm_update(C.byref(o), P, D, ptr(go, C.c_int32), ptr(occ, C.c_double),
                                         ptr(ma, C.c_double) if ma is not None else None,
                                         ptr(va, C.c_double) if va is not None else None,
                                         C.c_uint16(int(acc_flags)), C.c_uint16(int(flags)), ptr(w, C.c_float),
                                         ptr(gc, C.c_float), ptr(miv, C.c_float), ptr(iv, C.c_float),
                                         ptr(new_off, C.c_int32), C.byref(oc), C.byref(cnt), C.byref(fe),
                                         C.byref(fg), C.byref(rm)))
    n = int(new_off[-1])
    return new_off, w[:n], gc[:n], miv[:n], iv[:n], oc.value, cnt.value, fe.value, fg.value, rm.value


def mle_diag_gmm_update(config: MleDiagGmmOptions, diag_gmm_acc: AccumDiagGmm, flags, gmm: DiagGmm):
    """csrc/mle-diag-gmm.cc:243-390 -> (objf_change, count, floored_elements, floored_gaussians, removed)."""
    if gmm.num_gauss != diag_gmm_acc.num_gauss or gmm.dim != diag_gmm_acc.dim:
        raise KhgError("diag_gmm_acc.NumGauss() == gmm->NumGauss() assertion failed")
    r = _flat_update(config, [0, gmm.num_gauss], diag_gmm_acc.occupancy, diag_gmm_acc.mean_accumulator,
                     diag_gmm_acc.variance_accumulator, diag_gmm_acc.flags, int(flags), gmm._weights, gmm._means_invvars,
                     gmm._inv_vars)
    _, gmm._weights, gmm._gconsts, gmm._means_invvars, gmm._inv_vars = r[0], r[1].copy(), r[2].copy(), r[3].copy(), r[4].copy()
    gmm._valid_gconsts = True
    return r[5], r[6], r[7], r[8], r[9]


def ml_objective(gmm: DiagGmm, diaggmm_acc: AccumDiagGmm) -> float:   # csrc/mle-diag-gmm.cc:479-499
    obj = f32(np.dot(diaggmm_acc.occupancy, gmm._gconsts.astype(f64)))
    if diaggmm_acc.flags & 1:
        obj = f32(obj + (diaggmm_acc.mean_accumulator * gmm._means_invvars.astype(f64)).sum())
    if diaggmm_acc.flags & 2:
        obj = f32(obj - 0.5 * (diaggmm_acc.variance_accumulator * gmm._inv_vars.astype(f64)).sum())
    return float(obj)


class AccumAmDiagGmm:
    """csrc/mle-am-diag-gmm.h:18-97."""

    def __init__(self):
        self._accs: List[AccumDiagGmm] = []
        self._total_frames = 0.0
        self._total_log_like = 0.0

    def init(self, model: AmDiagGmm, dim_or_flags, flags=None):
        if flags is None:
            dim, flags = None, dim_or_flags
        else:
            dim = dim_or_flags
            if not dim > 0:
                raise KhgError("dim > 0 assertion failed")
        self._accs = []
        for i in range(model.num_pdfs):
            a = AccumDiagGmm()
            a.resize(model.get_pdf(i).num_gauss, dim if dim is not None else model.get_pdf(i).dim, flags)
            self._accs.append(a)

    def set_zero(self, flags):
        for a in self._accs:
            a.set_zero(flags)

    @property
    def num_accs(self):
        return len(self._accs)

    @property
    def dim(self):
        return self._accs[0].dim if self._accs else 0

    @property
    def tot_stats_count(self) -> float:
        return float(f32(sum(a.occupancy.sum() for a in self._accs)))

    @property
    def tot_count(self) -> float:      # float cast of the double (csrc/mle-am-diag-gmm.h:75)
        return float(f32(self._total_frames))

    @property
    def tot_log_like(self) -> float:   # :76
        return float(f32(self._total_log_like))

    def get_acc(self, index: int) -> AccumDiagGmm:   # the binding returns a COPY
        if not 0 <= index < len(self._accs):
            raise KhgError("index >= 0 && index < NumAccs() assertion failed")
        return self._accs[index].copy()

    def _chk(self, i):
        if not 0 <= i < len(self._accs):
            raise KhgError("gmm_index >= 0 && gmm_index < NumAccs() assertion failed")

    def accumulate_for_gmm(self, model: AmDiagGmm, data, gmm_index: int, weight: float) -> float:   # .cc:41-52
        self._chk(gmm_index)
        ll = self._accs[gmm_index].accumulate_from_diag(model.get_pdf(gmm_index), data, weight)
        self._total_log_like += float(f32(f32(ll) * f32(weight)))
        self._total_frames += float(f32(weight))
        return ll

    def accumulate_for_gmm_two_feats(self, model: AmDiagGmm, data1, data2, gmm_index: int, weight: float) -> float:
        """.cc:54-76: posteriors of data1 under the pdf, statistics of data2."""
        self._chk(gmm_index)
        ll, post = model.get_pdf(gmm_index).component_posteriors(data1)
        w = f32(weight)
        self._accs[gmm_index].accumulate_from_posteriors(data2, post * w)
        self._total_log_like += float(f32(f32(ll) * w))
        self._total_frames += float(w)
        return float(f32(ll))

    def accumulate_from_posteriors(self, model: AmDiagGmm, data, gmm_index: int, posteriors):
        self._chk(gmm_index)
        self._accs[gmm_index].accumulate_from_posteriors(data, posteriors)
        self._total_frames += float(np.asarray(posteriors, f32).sum(dtype=f32))

    def accumulate_for_gaussian(self, am: AmDiagGmm, data, gmm_index: int, gauss_index: int, weight: float):
        self._chk(gmm_index)
        if not 0 <= gauss_index < am.get_pdf(gmm_index).num_gauss:
            raise KhgError("gauss_index out of range")
        self._accs[gmm_index].accumulate_for_component(data, gauss_index, weight)

    def add(self, scale: float, other: "AccumAmDiagGmm"):   # .cc:119-128 == the cross-job sum / all-reduce
        s = float(f32(scale))
        self._total_frames += s * other._total_frames
        self._total_log_like += s * other._total_log_like
        if self.num_accs != other.num_accs:
            raise KhgError("num_accs == other.NumAccs() assertion failed")
        for a, b in zip(self._accs, other._accs):
            a.add(scale, b)

    def scale(self, scale: float):
        for a in self._accs:
            a.scale(scale, a.flags)
        self._total_frames *= float(f32(scale))
        self._total_log_like *= float(f32(scale))

    # ---- bridge to the device buffer (one contiguous fp64 block, see include/khg_hip.h) ----
    def add_device_stats(self, st: dict, gauss_off):
        """Add a downloaded DeviceAccs.split() dict (after any all-reduce) into these accumulators."""
        for i, a in enumerate(self._accs):
            lo, hi = int(gauss_off[i]), int(gauss_off[i + 1])
            a.occupancy += st["occ"][lo:hi]
            if a.flags & 1:
                a.mean_accumulator += st["mean_acc"][lo:hi]
            if a.flags & 2:
                a.variance_accumulator += st["var_acc"][lo:hi]
        self._total_frames += st["total_frames"]
        self._total_log_like += st["total_log_like"]


def mle_am_diag_gmm_update(config: MleDiagGmmOptions, amdiag_gmm_acc: AccumAmDiagGmm, flags, am_gmm: AmDiagGmm):
    """csrc/mle-am-diag-gmm.cc:153-202 -> (objf_change, count); am_gmm is updated in place."""
    if amdiag_gmm_acc.num_accs != am_gmm.num_pdfs:
        raise KhgError("am_diag_gmm_acc.NumAccs() == am_gmm->NumPdfs() assertion failed")
    if amdiag_gmm_acc.dim != am_gmm.dim:
        raise KhgError("accumulator / model dimension mismatch (ResizeModel path is not supported)")
    accs = amdiag_gmm_acc._accs
    acc_flags = accs[0].flags
    go = np.concatenate([[0], np.cumsum([p.num_gauss for p in am_gmm._pdfs])]).astype(np.int32)
    w = np.concatenate([p._weights for p in am_gmm._pdfs])
    miv = np.concatenate([p._means_invvars for p in am_gmm._pdfs])
    iv = np.concatenate([p._inv_vars for p in am_gmm._pdfs])
    occ = np.concatenate([a.occupancy for a in accs])
    ma = np.concatenate([a.mean_accumulator for a in accs]) if acc_flags & 1 else None
    va = np.concatenate([a.variance_accumulator for a in accs]) if acc_flags & 2 else None
    new_off, w, gc, miv, iv, oc, cnt, _, _, _ = _flat_update(config, go, occ, ma, va, acc_flags, int(flags), w, miv, iv)
    am_gmm.set_flat(new_off, w, gc, miv, iv)
    return oc, cnt


def mle_am_diag_gmm_update_device(config: MleDiagGmmOptions, device_accs, flags, device_model, am_gmm: AmDiagGmm = None):
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
