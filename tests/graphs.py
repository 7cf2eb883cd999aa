"""Random decoding graphs for the decoder tests: linear chains with optional-silence style
branches, epsilon-input arcs with word labels, unreachable finals."""
import numpy as np


def random_graph(rng, num_tids, n_main=8, p_branch=0.3, p_eps=0.2, with_final=True, p_long=0.0):
    """Left-to-right graph over states 0..n: every state gets a self-loop (tid) and a forward arc;
    some get a skip/branch arc, some an epsilon-input arc carrying a word label."""
    arcs = []   # (src, ilabel, olabel, weight, dst)
    n = n_main
    for s in range(n):
        tid_f = int(rng.integers(1, num_tids + 1))
        arcs.append((s, tid_f, 0, float(rng.random()), s + 1))
        if s > 0:
            arcs.append((s, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), s))      # self loop
        if s + 2 <= n and rng.random() < p_branch:
            arcs.append((s, int(rng.integers(1, num_tids + 1)), 0, float(rng.random() + 0.5), s + 2))
        for hop in (3, 4):                                   # longer skips: in-degrees of 4..6 (K2's <.,6,true> instantiation)
            if p_long and s + hop <= n and rng.random() < p_long:
                arcs.append((s, int(rng.integers(1, num_tids + 1)), 0, float(rng.random() + 0.5), s + hop))
        if s + 1 <= n and rng.random() < p_eps:
            arcs.append((s, 0, int(rng.integers(1, 50)), float(rng.random() * 0.3), s + 1))     # eps:word
    arcs.append((n, int(rng.integers(1, num_tids + 1)), 0, 0.1, n))
    S = n + 1
    arcs.sort(key=lambda a: a[0])
    arc_off = np.zeros(S + 1, np.int64)
    for a in arcs:
        arc_off[a[0] + 1] += 1
    arc_off = np.cumsum(arc_off)
    final = np.full(S, np.inf, np.float32)
    if with_final:
        final[n] = float(rng.random())
    return {
        "start": 0, "arc_off": arc_off,
        "ilabel": np.array([a[1] for a in arcs], np.int32), "olabel": np.array([a[2] for a in arcs], np.int32),
        "weight": np.array([a[3] for a in arcs], np.float32), "nextstate": np.array([a[4] for a in arcs], np.int32),
        "final": final,
    }


def concat(gs):
    out = {"state_off": [0], "start": [], "arc_off": [np.zeros(1, np.int64)]}
    for k in ("ilabel", "olabel", "weight", "nextstate", "final"):
        out[k] = []
    na = 0
    for g in gs:
        out["state_off"].append(out["state_off"][-1] + len(g["final"]))
        out["start"].append(g["start"])
        out["arc_off"].append(g["arc_off"][1:] + na)
        na += int(g["arc_off"][-1])
        for k in ("ilabel", "olabel", "weight", "nextstate", "final"):
            out[k].append(g[k])
    res = {k: np.concatenate(v) for k, v in out.items() if k not in ("state_off", "start")}
    res["state_off"] = np.asarray(out["state_off"], np.int64)
    res["start"] = np.asarray(out["start"], np.int32)
    return res
