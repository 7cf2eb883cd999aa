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


def permute_states(g, rng):
    """The same graph with its states renumbered at random (arc order inside a state kept): what a compiled graph's numbering
    looks like to a hash table keyed by the state id."""
    S = len(g["final"])
    perm = rng.permutation(S)                              # old -> new
    inv = np.argsort(perm)                                 # new -> old
    deg = np.diff(g["arc_off"])
    arc_off = np.concatenate([[0], np.cumsum(deg[inv])]).astype(np.int64)
    idx = np.concatenate([np.arange(g["arc_off"][o], g["arc_off"][o + 1]) for o in inv]) if S else np.zeros(0, np.int64)
    out = {"start": int(perm[g["start"]]), "arc_off": arc_off, "final": g["final"][inv]}
    for k in ("ilabel", "olabel", "weight"):
        out[k] = g[k][idx]
    out["nextstate"] = perm[g["nextstate"][idx]].astype(np.int32)
    return out


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


def hub_graph(rng, num_tids, fan=20, tail=6, eps_ties=True):
    """A start state fanning out to `fan` branches (out-degree far above the arcs of a training graph), every branch a
    short chain into a common tail; epsilon-input arcs in sequence and in parallel, some with EQUAL weights (which of two
    equal-cost epsilon paths sets the back-pointer, and where a state first reached through an epsilon arc enters the
    token list, depend on the reference's worklist order)."""
    arcs = []
    nxt = 1
    join = 1 + 2 * fan                                    # first state of the common tail
    for b in range(fan):
        s1, s2 = nxt, nxt + 1
        nxt += 2
        arcs.append((0, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), s1))
        arcs.append((s1, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), s1))
        arcs.append((s1, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), s2))
        arcs.append((s2, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), s2))
        w = 0.25 if eps_ties else float(rng.random() * 0.3)
        arcs.append((s2, 0, int(rng.integers(1, 50)), w, join))                      # eps:word into the tail (ties)
        if b % 3 == 0:
            arcs.append((s1, 0, 0, 0.125 if eps_ties else float(rng.random() * 0.3), s2))   # eps chain s1 -> s2 -> join
        arcs.append((s2, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), join))
    n = join + tail
    for s in range(join, n):
        arcs.append((s, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), s + 1))
        arcs.append((s, int(rng.integers(1, num_tids + 1)), 0, float(rng.random()), s))
        if s + 2 <= n:
            arcs.append((s, 0, int(rng.integers(1, 50)), 0.25 if eps_ties else float(rng.random() * 0.3), s + 2))
            arcs.append((s, 0, 0, 0.125, s + 1))                                     # with s+1 -eps(0.125)-> s+2: equal-cost pair
    arcs.append((n, int(rng.integers(1, num_tids + 1)), 0, 0.1, n))
    S = n + 1
    arcs.sort(key=lambda a: a[0])
    arc_off = np.zeros(S + 1, np.int64)
    for a in arcs:
        arc_off[a[0] + 1] += 1
    arc_off = np.cumsum(arc_off)
    final = np.full(S, np.inf, np.float32)
    final[n] = float(rng.random())
    return {
        "start": 0, "arc_off": arc_off,
        "ilabel": np.array([a[1] for a in arcs], np.int32), "olabel": np.array([a[2] for a in arcs], np.int32),
        "weight": np.array([a[3] for a in arcs], np.float32), "nextstate": np.array([a[4] for a in arcs], np.int32),
        "final": final,
    }
