"""DiagGmm / AmDiagGmm -- mirrors of csrc/diag-gmm.{h,cc}, csrc/am-diag-gmm.{h,cc} and their
pybind surface (python/csrc/diag-gmm.cc, am-diag-gmm.cc).  Parameters live in numpy (fp32, the
reference's exponential form); every likelihood / posterior evaluation runs on the GPU."""
import ctypes as C
from typing import List, Optional

import numpy as np

from . import _gpu, _lib
from ._lib import KhgError, check, lib, ptr

f32 = np.float32


class DiagGmm:
    def __init__(self, nmix: int = 0, dim: int = 0, gmm: "DiagGmm" = None):
        if gmm is not None:
            self.copy_from_diag_gmm(gmm)
        else:
            self.resize(nmix, dim)

    # ---- storage (csrc/diag-gmm.h:243-256) ----
    def resize(self, nmix: int, dim: int):
        self._gconsts = np.zeros(nmix, f32)
        self._weights = np.zeros(nmix, f32)
        self._inv_vars = np.ones((nmix, dim), f32)     # csrc/diag-gmm.cc:30-47 Resize: vars = 1
        self._means_invvars = np.zeros((nmix, dim), f32)
        self._valid_gconsts = False

    def copy_from_diag_gmm(self, o: "DiagGmm"):
        self._gconsts = o._gconsts.copy()
        self._weights = o._weights.copy()
        self._inv_vars = o._inv_vars.copy()
        self._means_invvars = o._means_invvars.copy()
        self._valid_gconsts = o._valid_gconsts

    @property
    def num_gauss(self) -> int:
        return self._weights.shape[0]

    @property
    def dim(self) -> int:
        return self._inv_vars.shape[1]

    @property
    def valid_gconsts(self) -> bool:
        return self._valid_gconsts

    @property
    def gconsts(self):
        return self._gconsts.copy()

    @property
    def weights(self):
        return self._weights.copy()

    @weights.setter
    def weights(self, w):
        self.set_weights(w)

    @property
    def means_invvars(self):
        return self._means_invvars.copy()

    @property
    def inv_vars(self):
        return self._inv_vars.copy()

    @property
    def means(self):      # GetMeans, csrc/diag-gmm.cc:956-958
        return self._means_invvars / self._inv_vars

    @property
    def vars(self):       # GetVars
        return (f32(1.0) / self._inv_vars).astype(f32)

    # ---- setters (csrc/diag-gmm.cc:940-1021) ----
    def set_weights(self, w):
        w = np.asarray(w, f32).reshape(-1)
        if w.shape[0] != self.num_gauss:
            raise KhgError("weights_.size() == w.size() assertion failed")
        self._weights = w.copy()
        self._valid_gconsts = False

    def set_means(self, m):
        m = np.asarray(m, f32)
        if m.shape != self._means_invvars.shape:
            raise KhgError("SetMeans: shape mismatch")
        self._means_invvars = (m * self._inv_vars).astype(f32)
        self._valid_gconsts = False

    def set_invvars(self, inv_vars):
        v = np.asarray(inv_vars, f32)
        if v.shape != self._inv_vars.shape:
            raise KhgError("SetInvVars: shape mismatch")
        self._means_invvars = (self._means_invvars / self._inv_vars * v).astype(f32)
        self._inv_vars = v.copy()
        self._valid_gconsts = False

    def set_invvars_and_means(self, invvars, means):
        v, m = np.asarray(invvars, f32), np.asarray(means, f32)
        if v.shape != self._inv_vars.shape or m.shape != v.shape:
            raise KhgError("SetInvVarsAndMeans: shape mismatch")
        self._inv_vars = v.copy()
        self._means_invvars = (m * v).astype(f32)
        self._valid_gconsts = False

    def set_component_weight(self, gauss: int, weight: float):
        if not weight > 0.0 or not gauss < self.num_gauss:
            raise KhgError("SetComponentWeight assertion failed")
        self._weights[gauss] = weight
        self._valid_gconsts = False

    def set_component_mean(self, gauss: int, v):
        self._means_invvars[gauss] = self._inv_vars[gauss] * np.asarray(v, f32)
        self._valid_gconsts = False

    def set_component_inv_var(self, gauss: int, v):
        v = np.asarray(v, f32)
        self._means_invvars[gauss] = self._means_invvars[gauss] / self._inv_vars[gauss] * v
        self._inv_vars[gauss] = v
        self._valid_gconsts = False

    def get_component_mean(self, gauss: int):
        return self._means_invvars[gauss] / self._inv_vars[gauss]

    def get_component_variance(self, gauss: int):
        return (f32(1.0) / self._inv_vars[gauss]).astype(f32)

    def remove_component(self, gauss: int, renorm_weights: bool):   # csrc/diag-gmm.cc:868-938
        if not 0 <= gauss < self.num_gauss:
            raise KhgError("RemoveComponent: index out of range")
        if self.num_gauss == 1:
            raise KhgError("Attempting to remove the only remaining component.")
        keep = [i for i in range(self.num_gauss) if i != gauss]
        self._weights = self._weights[keep]
        self._gconsts = self._gconsts[keep]
        self._means_invvars = self._means_invvars[keep]
        self._inv_vars = self._inv_vars[keep]
        if renorm_weights:
            self._weights = (self._weights / self._weights.sum(dtype=f32)).astype(f32)
            self._valid_gconsts = False

    def remove_components(self, gauss: List[int], renorm_weights: bool):   # :853-866
        g = sorted(gauss)
        if len(set(g)) != len(g):
            raise KhgError("IsSortedAndUniq(gauss) assertion failed")
        for i, x in enumerate(g):
            self.remove_component(x - i, renorm_weights)

    # ---- gconsts: host C++ (khg_compute_gconsts, csrc/diag-gmm.cc:103-147) ----
    def compute_gconsts(self) -> int:
        go = np.array([0, self.num_gauss], np.int32)
        nb = C.c_int32()
        check(lib.khg_compute_gconsts(1, self.dim, ptr(go, C.c_int32), ptr(self._weights, C.c_float),
                                      ptr(np.ascontiguousarray(self._inv_vars), C.c_float),
                                      ptr(np.ascontiguousarray(self._means_invvars), C.c_float),
                                      ptr(self._gconsts, C.c_float), C.byref(nb)))
        self._valid_gconsts = True
        return nb.value

    def _need_gconsts(self):
        if not self._valid_gconsts:
            raise KhgError("Must call ComputeGconsts() before computing likelihood")

    # ---- likelihoods (GPU, K1) ----
    def _as_model(self, per_component: bool):
        G = self.num_gauss
        go = np.arange(G + 1, dtype=np.int32) if per_component else np.array([0, G], np.int32)
        return go, self._gconsts, self._means_invvars, self._inv_vars

    def log_likelihood(self, data) -> float:       # csrc/diag-gmm.cc:150-165
        self._need_gconsts()
        data = np.asarray(data, f32).reshape(-1)
        if data.shape[0] != self.dim:
            raise KhgError(f"DiagGmm::LogLikelihoods, dimension mismatch {data.shape[0]} vs. {self.dim}")
        return float(_gpu.loglikes(*self._as_model(False), data, [0])[0, 0])

    def log_likelihoods(self, data) -> np.ndarray:  # :167-176 (each Gaussian as its own 1-component pdf)
        data = np.asarray(data, f32).reshape(-1)
        if data.shape[0] != self.dim:
            raise KhgError(f"DiagGmm::LogLikelihoods, dimension mismatch {data.shape[0]} vs. {self.dim}")
        return _gpu.loglikes(*self._as_model(True), data, np.arange(self.num_gauss))[:, 0].copy()

    def log_likelihoods_matrix(self, data) -> np.ndarray:  # :177-189 -> [N, G]
        data = np.asarray(data, f32)
        if data.ndim != 2 or data.shape[0] == 0:
            raise KhgError("data.rows() != 0 assertion failed")
        if data.shape[1] != self.dim:
            raise KhgError(f"DiagGmm::LogLikelihoods, dimension mismatch {data.shape[1]} vs. {self.dim}")
        return _gpu.loglikes(*self._as_model(True), data, np.arange(self.num_gauss)).T.copy()

    def log_likelihoods_preselect(self, data, indices) -> np.ndarray:  # :191-200
        return self.log_likelihoods(data)[np.asarray(indices, np.int64)]

    def component_log_likelihood(self, data, comp_id: int) -> float:
        if not 0 <= comp_id < self.num_gauss:
            raise KhgError("comp_id out of range")
        return float(self.log_likelihoods(data)[comp_id])

    def component_posteriors(self, data):           # :368-392 -> (log_like, posteriors)  (GPU, K3)
        self._need_gconsts()
        data = np.asarray(data, f32).reshape(-1)
        st = _gpu.acc_stats(*self._as_model(False), data, [0], 1.0)
        return st["total_log_like"], st["occ"].astype(f32)

    # ---- mixing up (csrc/diag-gmm.cc:780-851); `randn` injects the reference's RandnVector ----
    def split(self, target_components: int, perturb_factor: float, history: Optional[list] = None, randn=None):
        cur = self.num_gauss
        if target_components < cur or cur == 0:
            raise KhgError(f"Cannot split from {cur} to {target_components} components")
        if target_components == cur:
            return
        rng = randn or (lambda d: np.random.standard_normal(d).astype(f32))
        D = self.dim
        w = np.zeros(target_components, f32); w[:cur] = self._weights
        miv = np.zeros((target_components, D), f32); miv[:cur] = self._means_invvars
        iv = np.zeros((target_components, D), f32); iv[:cur] = self._inv_vars
        while cur < target_components:
            mx = int(np.argmax(w[:cur]))           # first maximum, like the strict '>' scan
            if history is not None:
                history.append(mx)
            w[mx] = w[mx] / f32(2)
            w[cur] = w[mx]
            rv = (np.asarray(rng(D), f32) * np.sqrt(iv[mx])).astype(f32)
            iv[cur] = iv[mx]
            miv[cur] = miv[mx] + rv * f32(perturb_factor)
            miv[mx] = miv[mx] - rv * f32(perturb_factor)
            cur += 1
        self._weights, self._means_invvars, self._inv_vars = w, miv, iv
        self._gconsts = np.zeros(target_components, f32)
        self.compute_gconsts()

    def merge(self, target_components: int) -> List[int]:
        """DiagGmm::Merge (csrc/diag-gmm.cc:557-759) through the library's host entry point khg_diag_gmm_merge; returns the
        merge history [kept_0, removed_0, kept_1, removed_1, ...] like python/csrc/diag-gmm.cc:79-85."""
        import ctypes as C
        from ._lib import check, lib, ptr
        G = C.c_int32(self.num_gauss)
        w = np.array(self._weights, f32); miv = np.array(self._means_invvars, f32); iv = np.array(self._inv_vars, f32)
        gc = np.zeros(self.num_gauss, f32)
        hist = np.zeros(2 * max(self.num_gauss, 1), np.int32)
        nh = C.c_int32()
        check(lib.khg_diag_gmm_merge(C.byref(G), self.dim, int(target_components), ptr(w, C.c_float), ptr(gc, C.c_float),
                                     ptr(miv, C.c_float), ptr(iv, C.c_float), ptr(hist, C.c_int32), C.byref(nh)))
        if G.value != self.num_gauss:
            g = G.value
            self._weights, self._means_invvars, self._inv_vars, self._gconsts = w[:g].copy(), miv[:g].copy(), iv[:g].copy(), gc[:g].copy()
            self._valid_gconsts = True
        return hist[: nh.value].tolist()

    def perturb(self, perturb_factor: float, randn=None):   # csrc/diag-gmm.cc:463-484
        rng = randn or (lambda shape: np.random.standard_normal(shape).astype(f32))
        rv = np.asarray(rng(self._means_invvars.shape), f32) * np.sqrt(self._inv_vars)
        self._means_invvars = (self._means_invvars + rv * f32(perturb_factor)).astype(f32)
        self.compute_gconsts()

    def generate(self, randn=None) -> np.ndarray:   # csrc/diag-gmm.cc:410-446
        """One sample.  Like the reference, the component is picked with `tot * Randn() * 0.99999` -- a NORMAL
        deviate, not a uniform one (so component 0 is chosen for every non-positive draw)."""
        rng = randn or (lambda shape: np.random.standard_normal(shape).astype(f32))
        tot = f32(self._weights.sum())
        if not tot > 0.0:
            raise KhgError("tot > 0.0 assertion failed")
        r = float(tot) * float(np.asarray(rng(1)).reshape(-1)[0]) * 0.99999
        i, acc, n = 0, 0.0, self.num_gauss
        while i < n and acc + float(self._weights[i]) < r:
            acc += float(self._weights[i])
            i += 1
        i = min(i, n - 1)
        t = self._inv_vars[i]
        return (self._means_invvars[i] / t + np.asarray(rng(self.dim), f32).reshape(-1) / np.sqrt(t)).astype(f32)

    def interpolate(self, rho: float, source: "DiagGmm", flags=0x7):   # csrc/diag-gmm.cc:460-484 (flags default kGmmAll)
        if self.num_gauss != source.num_gauss or self.dim != source.dim:
            raise KhgError("NumGauss() == source.NumGauss() && Dim() == source.Dim() assertion failed")
        flags = int(flags)
        # DiagGmmNormal of both (double), csrc/diag-gmm-normal.cc:14-20
        w = self._weights.astype(np.float64); wv = 1.0 / self._inv_vars.astype(np.float64); wm = self._means_invvars.astype(np.float64) * wv
        tw = source._weights.astype(np.float64); tv = 1.0 / source._inv_vars.astype(np.float64); tmn = source._means_invvars.astype(np.float64) * tv
        rho = float(f32(rho))
        if flags & 0x4:
            w = w * (1.0 - rho) + tw * rho
            w = w / w.sum()
        if flags & 0x1:
            wm = wm * (1.0 - rho) + tmn * rho
        if flags & 0x2:
            wv = wv * (1.0 - rho) + tv * rho
        # CopyToDiagGmm(kGmmAll) (csrc/diag-gmm-normal.cc:22-48)
        self._weights = w.astype(f32)
        self._inv_vars = (1.0 / wv).astype(f32)
        self._means_invvars = (wm.astype(f32) * self._inv_vars).astype(f32)
        self.compute_gconsts()

    # pickle: (weights, inv_vars, means_invvars); gconsts are re-derived (python/csrc/diag-gmm.cc:157-167)
    def __getstate__(self):
        return (self._weights, self._inv_vars, self._means_invvars)

    def __setstate__(self, t):
        self._weights = np.asarray(t[0], f32).copy()
        self._inv_vars = np.asarray(t[1], f32).copy()
        self._means_invvars = np.asarray(t[2], f32).copy()
        self._gconsts = np.zeros(self._weights.shape[0], f32)
        self.compute_gconsts()


class AmDiagGmm:
    """csrc/am-diag-gmm.h:96: one DiagGmm per pdf-id (ragged)."""

    def __init__(self):
        self._pdfs: List[DiagGmm] = []

    @property
    def dim(self) -> int:
        return self._pdfs[0].dim if self._pdfs else 0

    @property
    def num_pdfs(self) -> int:
        return len(self._pdfs)

    @property
    def num_gauss(self) -> int:
        return sum(p.num_gauss for p in self._pdfs)

    def num_gauss_in_pdf(self, pdf_index: int) -> int:
        return self.get_pdf(pdf_index).num_gauss

    def init(self, proto: DiagGmm, num_pdfs: int):
        self._pdfs = [DiagGmm(gmm=proto) for _ in range(num_pdfs)]

    def add_pdf(self, gmm: DiagGmm):
        if self._pdfs and gmm.dim != self.dim:
            raise KhgError("gmm.Dim() == this->Dim() assertion failed")
        self._pdfs.append(DiagGmm(gmm=gmm))

    def copy_from_am_diag_gmm(self, other: "AmDiagGmm"):
        self._pdfs = [DiagGmm(gmm=p) for p in other._pdfs]

    def get_pdf(self, pdf_index: int) -> DiagGmm:   # reference-returning, like the pybind binding
        if not 0 <= pdf_index < len(self._pdfs):
            raise KhgError("pdf_index out of range")
        return self._pdfs[pdf_index]

    def compute_gconsts(self) -> int:
        return sum(p.compute_gconsts() for p in self._pdfs)

    def log_likelihood(self, pdf_index: int, data) -> float:
        return self.get_pdf(pdf_index).log_likelihood(data)

    def get_gaussian_mean(self, pdf_index: int, gauss: int):
        return self.get_pdf(pdf_index).get_component_mean(gauss)

    def get_gaussian_variance(self, pdf_index: int, gauss: int):
        return self.get_pdf(pdf_index).get_component_variance(gauss)

    def set_gaussian_mean(self, pdf_index: int, gauss_index: int, v):
        self.get_pdf(pdf_index).set_component_mean(gauss_index, v)

    def split_pdf(self, pdf_idx: int, target_components: int, perturb_factor: float):
        self.get_pdf(pdf_idx).split(target_components, perturb_factor)

    def split_by_count(self, state_occs, target_components: int, perturb_factor: float, power: float,
                       min_count: float, randn=None):   # csrc/am-diag-gmm.cc:72-90
        from .mle import get_split_targets
        targets = get_split_targets(state_occs, target_components, power, min_count)
        for i, p in enumerate(self._pdfs):
            if p.num_gauss < targets[i]:
                p.split(targets[i], perturb_factor, randn=randn)

    def merge_by_count(self, state_occs, target_components: int, power: float, min_count: float):   # csrc/am-diag-gmm.cc:91-108
        from .mle import get_split_targets
        targets = get_split_targets(state_occs, target_components, power, min_count)
        for i, p in enumerate(self._pdfs):
            t = 1 if targets[i] == 0 else targets[i]      # can't merge below 1
            if p.num_gauss > t:
                p.merge(t)

    # ---- flat ragged view used by the device path ----
    def flat(self):
        go = np.concatenate([[0], np.cumsum([p.num_gauss for p in self._pdfs])]).astype(np.int32)
        for p in self._pdfs:
            p._need_gconsts()
        gc = np.concatenate([p._gconsts for p in self._pdfs]).astype(f32)
        w = np.concatenate([p._weights for p in self._pdfs]).astype(f32)
        miv = np.concatenate([p._means_invvars for p in self._pdfs]).astype(f32)
        iv = np.concatenate([p._inv_vars for p in self._pdfs]).astype(f32)
        return go, gc, w, miv, iv

    def set_flat(self, gauss_off, weights, gconsts, means_invvars, inv_vars):
        for i, p in enumerate(self._pdfs):
            a, b = int(gauss_off[i]), int(gauss_off[i + 1])
            p._weights = np.array(weights[a:b], f32)
            p._gconsts = np.array(gconsts[a:b], f32)
            p._means_invvars = np.array(means_invvars[a:b], f32)
            p._inv_vars = np.array(inv_vars[a:b], f32)
            p._valid_gconsts = True

    # pickle: flat tuple of 3*num_pdfs arrays (python/csrc/am-diag-gmm.cc:47-71)
    def __getstate__(self):
        out = []
        for p in self._pdfs:
            out += [p._weights, p._inv_vars, p._means_invvars]
        return tuple(out)

    def __setstate__(self, t):
        self._pdfs = []
        for i in range(0, len(t), 3):
            g = DiagGmm.__new__(DiagGmm)
            g.__setstate__((t[i], t[i + 1], t[i + 2]))
            self._pdfs.append(g)
