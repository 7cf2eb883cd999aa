"""Chain-form order-faithful decoder against the general wave form and the one-lane emulation on random epsilon-free graphs;
prints what distinguishes the first mismatching utterances."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth
from graphs import concat, random_graph
from oracle import oracle as orc
ctx = Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 101)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0 = time.time(); nb = nbad = nfb = 0
while time.time() - t0 < budget and nbad < 6:
    P = int(rng.choice([3, 6, 12, 30])); G = 1; D = int(rng.choice([2, 8, 13]))
    seed = int(rng.integers(1 << 30))
    m = synth.make_model(P, G, D, seed=seed)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    U = int(rng.integers(1, 14))
    p_br = float(rng.choice([0.0, 0.3, 0.8])); p_long = float(rng.choice([0.0, 0.5, 0.9]))
    graphs = [random_graph(rng, m.num_tids, n_main=int(rng.integers(1, 40)), p_branch=p_br, p_eps=0.0, with_final=True, p_long=p_long) for _ in range(U)]
    T = [int(rng.integers(max(1, len(g["final"]) - 2), len(g["final"]) + 40)) for g in graphs]
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    scale = float(rng.choice([0.1, 0.3, 1.0]))
    feats = (rng.standard_normal((int(frame_off[-1]), D)) * float(rng.choice([0.5, 3.0]))).astype(np.float32)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(np.zeros(m.num_tids + 1, np.float32))
    us = UtteranceSet(ctx, tm, frame_off, feats, graphs=concat(graphs))
    us.loglikes(dm)
    beam, retry = [(16.0, 0.0), (6.0, 40.0), (2.0, 8.0), (0.5, 1.0)][int(rng.integers(4))]
    kw = {}
    if rng.random() < 0.25:
        kw = {"max_active": int(rng.choice([2, 5, 30])), "min_active": int(rng.choice([0, 1]))}
    res = {}
    for mode in (0, 3, 1):
        ctx.set_option("k2_serial", mode)
        res[mode] = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=scale, **kw)
    ctx.set_option("k2_serial", 0)
    nb += 1
    nfb += int((res[0]["status"] & 8 != 0).sum())
    for u in range(U):
        a0, a3, a1 = (res[k]["ali"][frame_off[u]: frame_off[u + 1]] for k in (0, 3, 1))
        s0, s3, s1 = (int(res[k]["status"][u]) for k in (0, 3, 1))
        if s0 != s1 or not np.array_equal(a0, a1):
            nbad += 1
            g = graphs[u]
            deg = np.diff(g["arc_off"])
            d = np.nonzero(a0 != a1)[0]
            print(f"MISMATCH chain vs one-lane: P{P} D{D} U{U} u{u} S{len(g['final'])} T{T[u]} maxout{deg.max()} maxS_batch{max(len(x['final']) for x in graphs)} beam{beam}/{retry} scale{scale} {kw} "
                  f"status chain {s0} wave {s3} lane {s1}; wave==lane {np.array_equal(a3, a1)}; first diff frame {d[:3]} of {len(d)}; chain {a0[d[:3]]} lane {a1[d[:3]]}", flush=True)
    us.close(); tm.close(); dm.close()
print(f"{nb} batches, {nfb} fallback utterances, {nbad} mismatches")
