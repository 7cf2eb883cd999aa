"""ctypes binding of oracle/libkhg_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
See khg_oracle.h for the pinning status ("PARITY UNPINNED" for the decoder / M-step).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkhg_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("khg_oracle.c", "khg_oracle.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libkhg_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class Model(C.Structure):
    _fields_ = [("num_pdfs", C.c_int32), ("dim", C.c_int32), ("gauss_off", C.POINTER(C.c_int32)),
                ("gconsts", C.POINTER(C.c_float)), ("means_invvars", C.POINTER(C.c_float)),
                ("inv_vars", C.POINTER(C.c_float))]


class Graph(C.Structure):
    _fields_ = [("num_states", C.c_int32), ("start", C.c_int32), ("arc_off", C.POINTER(C.c_int32)),
                ("ilabel", C.POINTER(C.c_int32)), ("olabel", C.POINTER(C.c_int32)), ("weight", C.POINTER(C.c_float)),
                ("nextstate", C.POINTER(C.c_int32)), ("final", C.POINTER(C.c_float))]


class AlignConfig(C.Structure):
    _fields_ = [("beam", C.c_float), ("retry_beam", C.c_float), ("careful", C.c_int32), ("max_active", C.c_int32),
                ("min_active", C.c_int32), ("beam_delta", C.c_float), ("hash_ratio", C.c_float)]


class AlignStats(C.Structure):
    _fields_ = [("loglike_evals", C.c_int64), ("tokens_expanded", C.c_int64)]


class Accs(C.Structure):
    _fields_ = [("occ", C.POINTER(C.c_double)), ("mean_acc", C.POINTER(C.c_double)), ("var_acc", C.POINTER(C.c_double)),
                ("trans_acc", C.POINTER(C.c_double)), ("total_frames", C.c_double), ("total_log_like", C.c_double)]


class MleOpts(C.Structure):
    _fields_ = [("min_gaussian_weight", C.c_float), ("min_gaussian_occupancy", C.c_float), ("min_variance", C.c_double),
                ("remove_low_count_gaussians", C.c_int32), ("variance_floor_vector", C.POINTER(C.c_double))]


class OracleError(RuntimeError):
    pass


_lib = None
_variants = {}


def _open(path):
    l = C.CDLL(path)
    l.orc_logsumexp.restype = C.c_float
    l.orc_softmax.restype = C.c_float
    l.orc_ml_objective.restype = C.c_float
    l.orc_augment_gmm_flags.restype = C.c_uint16
    l.orc_augment_gmm_flags.argtypes = [C.c_uint16]
    l.orc_hl_create.restype = C.c_void_p
    l.orc_hl_destroy.argtypes = [C.c_void_p]
    l.orc_hl_set_size.argtypes = [C.c_void_p, C.c_int64]
    l.orc_hl_find.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    l.orc_hl_put.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    l.orc_hl_insert.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    l.orc_hl_list.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int64]
    l.orc_hl_list.restype = C.c_int64
    l.orc_hl_clear_reinsert.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    l.orc_hl_clear_reinsert.restype = C.c_int64
    l.orc_hl_drop.argtypes = [C.c_void_p]
    return l


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _variants[None] = _open(_LIB)
    return _lib


def use(variant=None):
    """Switch every wrapper in this module to another BUILD of the same source (bench.py's cpu_baseline):
    None = the checker (-O2), "o3" = -O3 without -march (the reference's default Release flags), "native" =
    -O3 -march=native (best-effort CPU).  Built on first use; not thread-safe, call it between timed phases."""
    global _lib
    if variant not in _variants:
        if variant is None:
            build()
            _variants[None] = _open(_LIB)
        else:
            name = f"libkhg_oracle_{variant}.so"
            # -march=native must be compiled on the machine that runs it (a copy built elsewhere travels with the
            # repository snapshot and may use instructions this CPU lacks): always rebuild that one
            cmd = ["make", "-C", _HERE] + (["-B"] if variant == "native" else []) + [name]
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
            _variants[variant] = _open(os.path.join(_HERE, name))
    _lib = _variants[variant]
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise OracleError(f"{what}: oracle error {rc}")


f32 = np.float32


class OModel:
    """Ragged AmDiagGmm for the oracle (keeps the numpy arrays alive)."""

    def __init__(self, gauss_off, gconsts, means_invvars, inv_vars):
        self.gauss_off = np.ascontiguousarray(gauss_off, np.int32)
        self.gconsts = np.ascontiguousarray(gconsts, f32)
        self.miv = np.ascontiguousarray(means_invvars, f32)
        self.iv = np.ascontiguousarray(inv_vars, f32)
        self.c = Model(self.gauss_off.shape[0] - 1, self.miv.shape[1], _p(self.gauss_off, C.c_int32),
                       _p(self.gconsts, C.c_float), _p(self.miv, C.c_float), _p(self.iv, C.c_float))


class OGraph:
    def __init__(self, start, arc_off, ilabel, olabel, weight, nextstate, final):
        self.arc_off = np.ascontiguousarray(arc_off, np.int32)
        self.ilabel = np.ascontiguousarray(ilabel, np.int32)
        self.olabel = np.ascontiguousarray(olabel, np.int32)
        self.weight = np.ascontiguousarray(weight, f32)
        self.nextstate = np.ascontiguousarray(nextstate, np.int32)
        self.final = np.ascontiguousarray(final, f32)
        self.c = Graph(self.final.shape[0], int(start), _p(self.arc_off, C.c_int32), _p(self.ilabel, C.c_int32),
                       _p(self.olabel, C.c_int32), _p(self.weight, C.c_float), _p(self.nextstate, C.c_int32),
                       _p(self.final, C.c_float))

    @staticmethod
    def from_set(graphs, u):
        """Slice utterance u out of the concatenated CSR used by the product."""
        s0, s1 = int(graphs["state_off"][u]), int(graphs["state_off"][u + 1])
        ao = graphs["arc_off"][s0: s1 + 1]
        a0, a1 = int(ao[0]), int(ao[-1])
        return OGraph(graphs["start"][u], ao - a0, graphs["ilabel"][a0:a1], graphs["olabel"][a0:a1],
                      graphs["weight"][a0:a1], graphs["nextstate"][a0:a1], graphs["final"][s0:s1])


def compute_gconsts(weights, inv_vars, means_invvars):
    w = np.ascontiguousarray(weights, f32); iv = np.ascontiguousarray(inv_vars, f32)
    miv = np.ascontiguousarray(means_invvars, f32)
    out = np.zeros(w.shape[0], f32)
    nb = C.c_int32()
    _chk(lib().orc_compute_gconsts(w.shape[0], iv.shape[1], _p(w, C.c_float), _p(iv, C.c_float), _p(miv, C.c_float),
                                   _p(out, C.c_float), C.byref(nb)), "compute_gconsts")
    return out, nb.value


def model_gconsts(gauss_off, weights, inv_vars, means_invvars):
    out = np.zeros(weights.shape[0], f32)
    for p in range(len(gauss_off) - 1):
        a, b = gauss_off[p], gauss_off[p + 1]
        out[a:b], _ = compute_gconsts(weights[a:b], inv_vars[a:b], means_invvars[a:b])
    return out


def loglikes(gconsts, means_invvars, inv_vars, x, fma_order=False):
    gc = np.ascontiguousarray(gconsts, f32); miv = np.ascontiguousarray(means_invvars, f32)
    iv = np.ascontiguousarray(inv_vars, f32); x = np.ascontiguousarray(x, f32)
    out = np.zeros(gc.shape[0], f32)
    fn = lib().orc_loglikes_fma_order if fma_order else lib().orc_loglikes
    fn(gc.shape[0], miv.shape[1], _p(gc, C.c_float), _p(miv, C.c_float), _p(iv, C.c_float), _p(x, C.c_float),
       _p(out, C.c_float))
    return out


def logsumexp(v):
    v = np.ascontiguousarray(v, f32)
    return float(lib().orc_logsumexp(v.shape[0], _p(v, C.c_float)))


def softmax(v):
    v = np.ascontiguousarray(v, f32)
    out = np.zeros_like(v)
    lse = lib().orc_softmax(v.shape[0], _p(v, C.c_float), _p(out, C.c_float))
    return out, float(lse)


def component_posteriors(m: OModel, pdf, x):
    x = np.ascontiguousarray(x, f32)
    G = int(m.gauss_off[pdf + 1] - m.gauss_off[pdf])
    post = np.zeros(G, f32)
    ll = C.c_float()
    _chk(lib().orc_component_posteriors(C.byref(m.c), int(pdf), _p(x, C.c_float), _p(post, C.c_float), C.byref(ll)),
         "component_posteriors")
    return post, ll.value


def loglikes_matrix(m: OModel, feats, pdfs):
    feats = np.ascontiguousarray(feats, f32); pdfs = np.ascontiguousarray(pdfs, np.int32)
    out = np.zeros((pdfs.shape[0], feats.shape[0]), f32)
    _chk(lib().orc_loglikes_matrix(C.byref(m.c), feats.shape[0], _p(feats, C.c_float), pdfs.shape[0],
                                   _p(pdfs, C.c_int32), _p(out, C.c_float)), "loglikes_matrix")
    return out


def add_transition_probs(ilabel, weight, log_probs, nsl, id2state, is_self_loop, transition_scale, self_loop_scale,
                         disambig=()):
    il = np.ascontiguousarray(ilabel, np.int32); w = np.array(weight, f32, copy=True)
    lp = np.ascontiguousarray(log_probs, f32); ns = np.ascontiguousarray(nsl, f32)
    i2s = np.ascontiguousarray(id2state, np.int32); sl = np.ascontiguousarray(is_self_loop, np.uint8)
    dis = np.ascontiguousarray(sorted(disambig), np.int32)
    _chk(lib().orc_add_transition_probs(il.shape[0], _p(il, C.c_int32), _p(w, C.c_float), lp.shape[0] - 1,
                                        _p(lp, C.c_float), _p(ns, C.c_float), _p(i2s, C.c_int32), _p(sl, C.c_uint8),
                                        C.c_float(transition_scale), C.c_float(self_loop_scale), dis.shape[0],
                                        _p(dis, C.c_int32)), "add_transition_probs")
    return w


def careful_graph(g: OGraph):
    S = g.final.shape[0]; A = g.ilabel.shape[0]
    ao = np.zeros(2 * S + 2, np.int32); il = np.zeros(2 * A + S + 1, np.int32); ol = np.zeros_like(il)
    w = np.zeros(2 * A + S + 1, f32); ns = np.zeros_like(il); fin = np.zeros(2 * S + 1, f32)
    nS = C.c_int32(); st = C.c_int32(); nA = C.c_int32()
    _chk(lib().orc_careful_graph(C.byref(g.c), C.byref(nS), C.byref(st), _p(ao, C.c_int32), _p(il, C.c_int32),
                                 _p(ol, C.c_int32), _p(w, C.c_float), _p(ns, C.c_int32), _p(fin, C.c_float),
                                 C.byref(nA)), "careful_graph")
    n = nA.value
    return OGraph(st.value, ao[: nS.value + 1], il[:n], ol[:n], w[:n], ns[:n], fin[: nS.value])


def _cfg(beam, retry_beam, max_active=2**31 - 1, min_active=20, beam_delta=0.5, hash_ratio=2.0):
    return AlignConfig(beam, retry_beam, 0, max_active, min_active, beam_delta, hash_ratio)


def align_utterance(g: OGraph, m: OModel, id2pdf, feats, acoustic_scale=1.0, beam=200.0, retry_beam=0.0, **kw):
    id2pdf = np.ascontiguousarray(id2pdf, np.int32); feats = np.ascontiguousarray(feats, f32)
    T = feats.shape[0]
    ali = np.zeros(max(T, 1), np.int32); words = np.zeros(T + g.final.shape[0] + 8, np.int32)
    nw = C.c_int32(); like = C.c_float(); status = C.c_int32(); st = AlignStats()
    cfg = _cfg(beam, retry_beam, **kw)
    rc = lib().orc_align_utterance(C.byref(cfg), C.c_float(acoustic_scale), C.byref(g.c), C.byref(m.c),
                                   _p(id2pdf, C.c_int32), id2pdf.shape[0] - 1, T, _p(feats, C.c_float),
                                   _p(ali, C.c_int32), _p(words, C.c_int32), words.shape[0], C.byref(nw),
                                   C.byref(like), C.byref(status), C.byref(st))
    _chk(rc, "align_utterance")
    ok = (status.value & 1) == 0
    return {"ali": ali[:T] if ok else np.zeros(0, np.int32), "words": words[: nw.value], "like": like.value,
            "status": status.value, "loglike_evals": st.loglike_evals, "tokens_expanded": st.tokens_expanded}


def align_utterance_ll(g: OGraph, id2pdf, T, pdfs, ll, acoustic_scale=1.0, beam=200.0, retry_beam=0.0, **kw):
    id2pdf = np.ascontiguousarray(id2pdf, np.int32); pdfs = np.ascontiguousarray(pdfs, np.int32)
    ll = np.ascontiguousarray(ll, f32)
    ali = np.zeros(max(T, 1), np.int32); words = np.zeros(T + g.final.shape[0] + 8, np.int32)
    nw = C.c_int32(); like = C.c_float(); status = C.c_int32(); st = AlignStats()
    cfg = _cfg(beam, retry_beam, **kw)
    rc = lib().orc_align_utterance_ll(C.byref(cfg), C.c_float(acoustic_scale), C.byref(g.c), _p(id2pdf, C.c_int32),
                                      id2pdf.shape[0] - 1, T, pdfs.shape[0], _p(pdfs, C.c_int32), _p(ll, C.c_float),
                                      C.c_int64(ll.shape[1] if ll.ndim == 2 else T), _p(ali, C.c_int32),
                                      _p(words, C.c_int32), words.shape[0], C.byref(nw), C.byref(like),
                                      C.byref(status), C.byref(st))
    _chk(rc, "align_utterance_ll")
    ok = (status.value & 1) == 0
    return {"ali": ali[:T] if ok else np.zeros(0, np.int32), "words": words[: nw.value], "like": like.value,
            "status": status.value}


def exact_viterbi_ll(g: OGraph, id2pdf, T, pdfs, ll, acoustic_scale=1.0):
    id2pdf = np.ascontiguousarray(id2pdf, np.int32); pdfs = np.ascontiguousarray(pdfs, np.int32)
    ll = np.ascontiguousarray(ll, f32)
    ali = np.zeros(max(T, 1), np.int32); bc = C.c_double(); status = C.c_int32()
    rc = lib().orc_exact_viterbi_ll(C.c_float(acoustic_scale), C.byref(g.c), _p(id2pdf, C.c_int32),
                                    id2pdf.shape[0] - 1, T, pdfs.shape[0], _p(pdfs, C.c_int32), _p(ll, C.c_float),
                                    C.c_int64(ll.shape[1] if ll.ndim == 2 else T), _p(ali, C.c_int32), C.byref(bc),
                                    C.byref(status))
    _chk(rc, "exact_viterbi_ll")
    return {"ali": ali[:T], "cost": bc.value, "status": status.value}


class OAccs:
    def __init__(self, sumG, D, num_tids):
        self.occ = np.zeros(sumG, np.float64); self.mean_acc = np.zeros((sumG, D), np.float64)
        self.var_acc = np.zeros((sumG, D), np.float64); self.trans_acc = np.zeros(num_tids + 1, np.float64)
        self.c = Accs(_p(self.occ, C.c_double), _p(self.mean_acc, C.c_double), _p(self.var_acc, C.c_double),
                      _p(self.trans_acc, C.c_double), 0.0, 0.0)

    @property
    def total_frames(self):
        return self.c.total_frames

    @property
    def total_log_like(self):
        return self.c.total_log_like


def acc_stats_ali(m: OModel, id2pdf, feats, ali, accs: OAccs, weight=1.0):
    id2pdf = np.ascontiguousarray(id2pdf, np.int32); feats = np.ascontiguousarray(feats, f32)
    ali = np.ascontiguousarray(ali, np.int32)
    ll = C.c_double()
    _chk(lib().orc_acc_stats_ali(C.byref(m.c), _p(id2pdf, C.c_int32), id2pdf.shape[0] - 1, feats.shape[0],
                                 _p(feats, C.c_float), _p(ali, C.c_int32), C.c_float(weight), C.byref(accs.c),
                                 C.byref(ll)), "acc_stats_ali")
    return ll.value


def em_pass_mt(m: "OModel", id2pdf, graphs: dict, frame_off, feats, first_utt=0, n_utt=None, num_threads=1, budget_seconds=1e9,
               acoustic_scale=1.0, beam=200.0, retry_beam=0.0, keep=None, **kw):
    """orc_em_pass_mt: align + acc-stats per utterance on `num_threads` POSIX threads (bench.py's cpu_baseline, variant B).
    `graphs` is the concatenated CSR dict of the utterance set (weights already carrying the transition costs).
    -> (frames_done, utterances_done, failed, seconds).  With keep = a dict, what the pass computed is left in it: "ali" (int32 per
    frame of the utterances handed in, 0 where an utterance failed or was not reached), "status" (-1 = not reached; the reached
    utterances are a prefix), "like", and "accs" (an OAccs: the threads' accumulators summed)."""
    id2pdf = np.ascontiguousarray(id2pdf, np.int32)
    fo = np.ascontiguousarray(frame_off, np.int64)
    x = np.ascontiguousarray(feats, f32)
    g = {k: np.ascontiguousarray(graphs[k], dt) for k, dt in (("state_off", np.int64), ("start", np.int32), ("arc_off", np.int64),
                                                                ("ilabel", np.int32), ("olabel", np.int32), ("weight", f32),
                                                                ("nextstate", np.int32), ("final", f32))}
    n_all = fo.shape[0] - 1
    n_utt = n_all - first_utt if n_utt is None else min(n_utt, n_all - first_utt)
    cfg = _cfg(beam, retry_beam, **kw)
    frames = C.c_int64(); utts = C.c_int32(); failed = C.c_int32(); secs = C.c_double()
    args = (C.byref(cfg), C.c_float(acoustic_scale), C.byref(m.c), _p(id2pdf, C.c_int32), id2pdf.shape[0] - 1,
            C.c_int32(first_utt), C.c_int32(max(n_utt, 0)), _p(fo, C.c_int64), _p(x, C.c_float),
            _p(g["state_off"], C.c_int64), _p(g["start"], C.c_int32), _p(g["arc_off"], C.c_int64),
            _p(g["ilabel"], C.c_int32), _p(g["olabel"], C.c_int32), _p(g["weight"], C.c_float),
            _p(g["nextstate"], C.c_int32), _p(g["final"], C.c_float), C.c_int32(num_threads),
            C.c_double(budget_seconds), C.byref(frames), C.byref(utts), C.byref(failed), C.byref(secs))
    if keep is None:
        _chk(lib().orc_em_pass_mt(*args), "em_pass_mt")
    else:
        n = max(n_utt, 0)
        keep["ali"] = np.zeros(max(int(fo[first_utt + n] - fo[first_utt]), 1), np.int32)
        keep["status"] = np.full(max(n, 1), -1, np.int32)
        keep["like"] = np.zeros(max(n, 1), f32)
        keep["accs"] = OAccs(int(m.gauss_off[-1]), m.miv.shape[1], id2pdf.shape[0] - 1)
        fn = lib().orc_em_pass_mt_keep
        fn.restype = C.c_int
        _chk(fn(*args, _p(keep["ali"], C.c_int32), _p(keep["status"], C.c_int32), _p(keep["like"], C.c_float), C.byref(keep["accs"].c)),
             "em_pass_mt_keep")
        keep["status"] = keep["status"][:n]; keep["like"] = keep["like"][:n]
    return frames.value, utts.value, failed.value, secs.value


def diag_gmm_merge(weights, means_invvars, inv_vars, target_components):
    """diag-gmm.cc:557-759 -> dict(weights, gconsts, means_invvars, inv_vars, history)."""
    w = np.array(weights, f32, copy=True); miv = np.array(means_invvars, f32, copy=True)
    iv = np.array(inv_vars, f32, copy=True); gc = np.zeros_like(w)
    G = C.c_int32(w.shape[0]); nh = C.c_int32()
    hist = np.zeros(2 * max(w.shape[0], 1), np.int32)
    _chk(lib().orc_diag_gmm_merge(C.byref(G), miv.shape[1], int(target_components), _p(w, C.c_float), _p(gc, C.c_float),
                                  _p(miv, C.c_float), _p(iv, C.c_float), _p(hist, C.c_int32), C.byref(nh)), "diag_gmm_merge")
    g = G.value
    return {"weights": w[:g], "gconsts": gc[:g], "means_invvars": miv[:g], "inv_vars": iv[:g], "history": hist[: nh.value].tolist()}


def mle_diag_gmm_update(weights, means_invvars, inv_vars, occ, mean_acc, var_acc, acc_flags=0xF, flags=0x7,
                        min_gaussian_weight=1e-5, min_gaussian_occupancy=10.0, min_variance=1e-3, remove=True,
                        variance_floor_vector=None):
    w = np.array(weights, f32, copy=True); miv = np.array(means_invvars, f32, copy=True)
    iv = np.array(inv_vars, f32, copy=True); gc = np.zeros_like(w)
    occ = np.ascontiguousarray(occ, np.float64); ma = np.ascontiguousarray(mean_acc, np.float64)
    va = np.ascontiguousarray(var_acc, np.float64)
    vfv = None if variance_floor_vector is None else np.ascontiguousarray(variance_floor_vector, np.float64)
    o = MleOpts(min_gaussian_weight, min_gaussian_occupancy, min_variance, int(remove), None if vfv is None else _p(vfv, C.c_double))
    G = C.c_int32(w.shape[0]); oc = C.c_float(); cnt = C.c_float(); fe = C.c_int32(); fg = C.c_int32(); rm = C.c_int32()
    _chk(lib().orc_mle_diag_gmm_update(C.byref(o), C.byref(G), miv.shape[1], _p(occ, C.c_double), _p(ma, C.c_double),
                                       _p(va, C.c_double), C.c_uint16(acc_flags), C.c_uint16(flags), _p(w, C.c_float),
                                       _p(gc, C.c_float), _p(miv, C.c_float), _p(iv, C.c_float), C.byref(oc),
                                       C.byref(cnt), C.byref(fe), C.byref(fg), C.byref(rm)), "mle_diag_gmm_update")
    g = G.value
    return {"weights": w[:g], "gconsts": gc[:g], "means_invvars": miv[:g], "inv_vars": iv[:g], "obj_change": oc.value,
            "count": cnt.value, "floored_elems": fe.value, "floored_gauss": fg.value, "removed": rm.value}


def map_diag_gmm_update(weights, means_invvars, inv_vars, occ, mean_acc, var_acc, flags=0x7, mean_tau=10.0, variance_tau=50.0,
                        weight_tau=10.0):
    """MapDiagGmmUpdate (csrc/mle-diag-gmm.cc:392-477) with DiagGmmNormal (csrc/diag-gmm-normal.cc:14-48) restated over arrays, one
    numpy fp64 statement per statement of the reference; taus are the reference's floats widened.  -> new (weights, means_invvars,
    inv_vars, gconsts, count).  (Test infrastructure; parity unpinned: the reference's tests hold no MAP answer, only the options'
    defaults, python/tests/test_mle_diag_gmm.py:34-46.)"""
    w32 = np.array(weights, f32, copy=True); miv32 = np.array(means_invvars, f32, copy=True); iv32 = np.array(inv_vars, f32, copy=True)
    occ = np.asarray(occ, np.float64); ma = np.asarray(mean_acc, np.float64); va = np.asarray(var_acc, np.float64)
    mean_tau, variance_tau, weight_tau = float(f32(mean_tau)), float(f32(variance_tau)), float(f32(weight_tau))
    occ_sum = 0.0
    for o in occ:
        occ_sum += float(o)
    w = w32.astype(np.float64); vars_ = 1.0 / iv32.astype(np.float64); means = miv32.astype(np.float64) * vars_
    old_means = means.copy()
    for i in range(w.shape[0]):
        o = float(occ[i])
        w[i] = (o + w[i] * weight_tau) / (occ_sum + weight_tau)
        if o > 0.0 and flags & 1:
            mean = ma[i] * (1.0 / (o + mean_tau))
            mean = mean + means[i] * (mean_tau / (o + mean_tau))
            means[i] = mean
        if o > 0.0 and flags & 2:
            old_var = vars_[i].copy()
            var = va[i] / o
            var = var + means[i] * means[i]
            var = var + ma[i] * means[i] * (-2.0 / o)
            var = var * (o / (variance_tau + o))
            var = var + old_var * (variance_tau / (variance_tau + o))
            vars_[i] = var
    if flags & 4:
        w32 = w.astype(f32)
    if flags & 2:
        iv32 = (1.0 / vars_).astype(f32)
        if not flags & 1:
            miv32 = (old_means.astype(f32) * iv32).astype(f32)
    if flags & 1:
        miv32 = (means.astype(f32) * iv32).astype(f32)
    gc, _ = compute_gconsts(w32, iv32, miv32)
    return {"weights": w32, "means_invvars": miv32, "inv_vars": iv32, "gconsts": gc, "count": float(f32(occ_sum))}


def transition_mle_update(state2id, self_loop_of, stats, log_probs, nsl, floor=0.01, mincount=5.0):
    s2i = np.ascontiguousarray(state2id, np.int32); slo = np.ascontiguousarray(self_loop_of, np.int32)
    st = np.ascontiguousarray(stats, np.float64); lp = np.array(log_probs, f32, copy=True); ns = np.array(nsl, f32, copy=True)
    oi = C.c_float(); cnt = C.c_float()
    _chk(lib().orc_transition_mle_update(slo.shape[0] - 1, _p(s2i, C.c_int32), _p(slo, C.c_int32), _p(st, C.c_double),
                                         C.c_float(floor), C.c_float(mincount), _p(lp, C.c_float), _p(ns, C.c_float),
                                         C.byref(oi), C.byref(cnt)), "transition_mle_update")
    return lp, ns, oi.value, cnt.value


def transition_mle_update_shared(state2id, fwd_pdf, stats, log_probs, floor=0.01, mincount=5.0):
    """TransitionModel::MleUpdateShared (transition-model.cc:531-655) for is_hmm topologies, restated over arrays:
    state2id[1..S+1] first transition-id of each transition-state, fwd_pdf[1..S] its pdf.  -> (new log_probs, objf_impr, count).
    (Test infrastructure; the non-self-loop log-probs are re-derived by the caller, transition-model.cc:339-359.)"""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype = ctypes.c_float; libm.logf.argtypes = [ctypes.c_float]
    libm.expf.restype = ctypes.c_float; libm.expf.argtypes = [ctypes.c_float]
    s2i = np.asarray(state2id, np.int64)
    S = len(s2i) - 2
    lp = np.array(log_probs, f32, copy=True)
    st = np.asarray(stats, np.float64)
    by_pdf = {}
    for ts in range(1, S + 1):
        by_pdf.setdefault(int(fwd_pdf[ts]), []).append(ts)
    count_sum = f32(0); objf = f32(0)
    for pdf in sorted(by_pdf):
        tss = by_pdf[pdf]
        n = int(s2i[tss[0] + 1] - s2i[tss[0]])
        if n <= 1:
            continue
        if any(int(s2i[t + 1] - s2i[t]) != n for t in tss):
            raise RuntimeError("Mismatch in #transition indices")
        rows = np.stack([st[s2i[t]: s2i[t] + n] for t in tss])            # [tstates][n]
        counts = np.zeros(n, np.float64); tot = 0.0
        for r in rows:                                                      # the reference's accumulation order
            for k in range(n):
                counts[k] += r[k]; tot += r[k]
        count_sum = f32(np.float64(count_sum) + tot)
        if tot < mincount:
            continue
        old = np.array([libm.expf(float(lp[s2i[tss[0]] + k])) for k in range(n)], f32)
        new = (counts / tot).astype(f32)
        for _ in range(3):
            ssum = f32(0)
            for x in new:
                ssum = f32(ssum + x)
            new = np.maximum((new / ssum).astype(f32), f32(floor))
        for k in range(n):
            d = f32(f32(libm.logf(float(new[k]))) - f32(libm.logf(float(old[k]))))
            objf = f32(np.float64(objf) + counts[k] * np.float64(d))
        for t in tss:
            for k in range(n):
                lp[s2i[t] + k] = libm.logf(float(new[k]))
    return lp, float(objf), float(count_sum)


def augment_gmm_flags(flags):
    return int(lib().orc_augment_gmm_flags(flags))


class OHashList:
    """The decoder's HashList restatement (hash-list.h, hash-list-inl.h) through the oracle's test hooks."""

    def __init__(self):
        self.h = lib().orc_hl_create()

    def set_size(self, n):
        lib().orc_hl_set_size(self.h, int(n))

    def find(self, key):
        v = C.c_int64()
        return int(v.value) if lib().orc_hl_find(self.h, int(key), C.byref(v)) else None

    def put(self, key, val):
        lib().orc_hl_put(self.h, int(key), int(val))

    def insert(self, key, val):
        return bool(lib().orc_hl_insert(self.h, int(key), int(val)))

    def items(self):
        n = lib().orc_hl_list(self.h, None, None, 0)
        k = np.zeros(max(n, 1), np.int32)
        v = np.zeros(max(n, 1), np.int64)
        lib().orc_hl_list(self.h, _p(k, C.c_int32), _p(v, C.c_int64), n)
        return list(zip(k[:n].tolist(), v[:n].tolist()))

    def clear_reinsert(self, new_size, shift=1):
        return int(lib().orc_hl_clear_reinsert(self.h, int(new_size), int(shift)))

    def drop(self):
        lib().orc_hl_drop(self.h)

    def close(self):
        if self.h:
            lib().orc_hl_destroy(self.h)
            self.h = None
