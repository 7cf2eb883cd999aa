"""Early-iteration behaviour: align with a MISMATCHED model (1 Gaussian per pdf at the global mean + noise, like the
flat start of a recipe) at beam 6 / retry 40 and report how many utterances need the serial order-faithful decoder
and what the alignment pass costs."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
P, G, D = 5000, 64, 40
U = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
m = synth.make_model(P, G, D, seed=1)
ut = synth.make_utts(m, U, seed=3)
rng = np.random.default_rng(0)
# flat-start-like model: every pdf = the same broad Gaussian, means perturbed a little
mean = ut.feats.mean(0); var = ut.feats.var(0)
go = np.arange(P + 1, dtype=np.int32)
means = (mean[None, :] + noise * np.sqrt(var)[None, :] * rng.standard_normal((P, D))).astype(np.float32)
iv = np.tile((1.0 / var).astype(np.float32), (P, 1))
miv = (means * iv).astype(np.float32)
w = np.ones(P, np.float32)
gc = np.zeros(P, np.float32)
_lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(go, C.c_int32), _lib.ptr(w, C.c_float), _lib.ptr(iv, C.c_float), _lib.ptr(miv, C.c_float), _lib.ptr(gc, C.c_float), None))
ctx = Context(0)
dm = DeviceModel(ctx, go, gc, miv, iv)
tm = DeviceTransitions(ctx, m.id2pdf)
cost = np.zeros(m.num_tids + 1, np.float32)
_lib.check(_lib.lib.khg_scaled_trans_cost(m.num_tids, _lib.ptr(m.log_probs, C.c_float), _lib.ptr(m.non_self_loop_log_probs, C.c_float), _lib.ptr(m.id2state, C.c_int32), _lib.ptr(m.is_self_loop, C.c_uint8), 1.0, 0.1, _lib.ptr(cost, C.c_float)))
tm.set_trans_cost(cost)
us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
us.loglikes(dm, reachable_only=True); ctx.sync()
for beam, retry in [(200, 0), (10, 40), (6, 40), (200, 0)]:
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
    ctx.set_timing(True)
    t0 = time.time()
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
    dt = time.time() - t0
    km = dict(ctx.timings()); ctx.set_timing(False)
    st = np.asarray(res["status"])
    print(f"beam {beam}/{retry}: {dt*1e3:.1f} ms for {U} utts  exact_dp={(st & 4 > 0).sum()} fallback={(st & 8 > 0).sum()} retried={(st & 2 > 0).sum()} error={(st & 1).sum()}  kernels " +
          ", ".join(f"{k} {v:.2f}" for k, v in km.items() if k.startswith("k2")))

# --- the same set on graphs WITH epsilon-input arcs (no beam certificate: the order-faithful decoder is their decoder) ---
# every phone-boundary state gets an epsilon arc that skips one HMM state at a price; the wave form (exact slot prefix sums,
# lane-0 ProcessNonemitting over the epsilon-capable tokens) against the one-lane emulation and the epsilon-free numbers above
def with_eps(g):
    so, ao = g["state_off"], g["arc_off"]
    NS = int(so[-1])
    local = np.arange(NS) - np.repeat(so[:-1], np.diff(so))
    last = np.repeat(np.diff(so) - 1, np.diff(so))
    add = (local % 3 == 0) & (local + 1 <= last)
    nar = np.diff(ao) + add
    ao2 = np.concatenate([[0], np.cumsum(nar)]).astype(np.int64)
    NA2 = int(ao2[-1])
    out = {k: np.zeros(NA2, g[k].dtype) for k in ("ilabel", "olabel", "weight", "nextstate")}
    # old arcs keep their order, the epsilon arc goes last in its state
    old_pos = np.arange(int(ao[-1])) + np.repeat(ao2[:-1] - ao[:-1], np.diff(ao))
    for k in out:
        out[k][old_pos] = g[k]
    ep = ao2[1:][add] - 1
    out["ilabel"][ep] = 0; out["olabel"][ep] = 7; out["weight"][ep] = 2.0; out["nextstate"][ep] = local[add] + 1
    return dict(g, arc_off=ao2, **out)

use = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=with_eps(ut.graphs))
use.loglikes(dm, reachable_only=True); ctx.sync()
for serial in (0, 1):
    ctx.set_option("k2_serial", serial)
    for beam, retry in [(6, 40), (200, 0)]:
        res = use.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
        ctx.set_timing(True)
        t0 = time.time()
        res = use.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1); ctx.sync()
        dt = time.time() - t0
        km = dict(ctx.timings()); ctx.set_timing(False)
        st = np.asarray(res["status"])
        print(f"eps graphs, {'one-lane' if serial else 'wave'} decoder, beam {beam}/{retry}: {dt*1e3:.1f} ms for {U} utts  fallback={(st & 8 > 0).sum()} retried={(st & 2 > 0).sum()} error={(st & 1).sum()}  kernels " +
              ", ".join(f"{k} {v:.2f}" for k, v in km.items() if k.startswith("k2")))
        if serial == 0: keep = res
        else: print("   identical to the wave form:", bool(np.array_equal(keep["ali"], res["ali"]) and np.array_equal(keep["status"], res["status"])) if (beam, retry) == (200, 0) else "-")
ctx.set_option("k2_serial", 0)
