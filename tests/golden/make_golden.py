#!/usr/bin/env python3
"""Generates tests/golden/em_small.npz with the CPU oracle (oracle/khg_oracle.c).

The reference cannot be imported or built in this environment, so these vectors are
outputs of the line-by-line restatement, not of the reference binary: they freeze the oracle's
behaviour (regression pin) and give the GPU tests inputs/expected outputs that travel to the GPU
box.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from graphs import concat, random_graph  # noqa: E402
from kaldi_hmm_gmm_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    rng = np.random.default_rng(20230414)
    m = synth.make_model(18, 4, 13, seed=20230414, ragged=True)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    om = orc.OModel(m.gauss_off, gc, m.means_invvars, m.inv_vars)
    U = 6
    graphs = [random_graph(rng, m.num_tids, n_main=int(rng.integers(4, 9))) for _ in range(U)]
    T = [int(rng.integers(len(g["final"]) + 2, 40)) for g in graphs]
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    feats = (rng.standard_normal((frame_off[-1], 13)) * 2.5).astype(np.float32)
    G = concat(graphs)
    cost = orc.add_transition_probs(np.arange(m.num_tids + 1, dtype=np.int32), np.zeros(m.num_tids + 1, np.float32),
                                    m.log_probs, m.non_self_loop_log_probs, m.id2state, m.is_self_loop, 1.0, 0.1)
    out = {"frame_off": frame_off, "feats": feats, "gauss_off": m.gauss_off, "weights": m.weights, "gconsts": gc,
           "means_invvars": m.means_invvars, "inv_vars": m.inv_vars, "id2pdf": m.id2pdf, "trans_cost": cost}
    for k, v in G.items():
        out["g_" + k] = v
    for tag, beam, retry in (("wide", 200.0, 0.0), ("narrow", 2.0, 6.0)):
        ali = np.zeros(frame_off[-1], np.int32); like = np.zeros(U, np.float32); status = np.zeros(U, np.int32)
        words = []
        for u in range(U):
            g = dict(graphs[u]); il = g["ilabel"]
            g["weight"] = np.where(il >= 1, g["weight"] + cost[il], g["weight"]).astype(np.float32)
            og = orc.OGraph(g["start"], g["arc_off"], g["ilabel"], g["olabel"], g["weight"], g["nextstate"], g["final"])
            r = orc.align_utterance(og, om, m.id2pdf, feats[frame_off[u]: frame_off[u + 1]], acoustic_scale=0.1, beam=beam,
                                    retry_beam=retry)
            status[u] = r["status"]; like[u] = r["like"]
            if (r["status"] & 1) == 0:
                ali[frame_off[u]: frame_off[u + 1]] = r["ali"]
            words.append(r["words"])
        out[f"ali_{tag}"] = ali; out[f"like_{tag}"] = like; out[f"status_{tag}"] = status
        out[f"words_{tag}"] = np.concatenate(words).astype(np.int32)
        out[f"words_off_{tag}"] = np.concatenate([[0], np.cumsum([len(w) for w in words])]).astype(np.int64)
    acc = orc.OAccs(int(m.gauss_off[-1]), 13, m.num_tids)
    tot = 0.0
    for u in range(U):
        sl = slice(frame_off[u], frame_off[u + 1])
        if out["status_wide"][u] & 1:
            continue
        tot += orc.acc_stats_ali(om, m.id2pdf, feats[sl], out["ali_wide"][sl], acc)
    out.update(occ=acc.occ, mean_acc=acc.mean_acc, var_acc=acc.var_acc, trans_acc=acc.trans_acc,
               total_frames=acc.total_frames, total_log_like=acc.total_log_like, acc_log_like=tot)
    # M-step on those statistics (mixup == current #Gauss: no random split), min occupancy 3 as the first yesno update
    new_w, new_gc, new_miv, new_iv, new_off = [], [], [], [], [0]
    objf = np.float32(0); cnt = np.float32(0)
    for p in range(18):
        a, b = m.gauss_off[p], m.gauss_off[p + 1]
        r = orc.mle_diag_gmm_update(m.weights[a:b], m.means_invvars[a:b], m.inv_vars[a:b], acc.occ[a:b], acc.mean_acc[a:b],
                                    acc.var_acc[a:b], acc_flags=0xF, flags=0x7, min_gaussian_occupancy=3.0)
        new_w.append(r["weights"]); new_gc.append(r["gconsts"]); new_miv.append(r["means_invvars"]); new_iv.append(r["inv_vars"])
        new_off.append(new_off[-1] + len(r["weights"]))
        objf = np.float32(objf + np.float32(r["obj_change"])); cnt = np.float32(cnt + np.float32(r["count"]))
    out.update(new_gauss_off=np.asarray(new_off, np.int32), new_weights=np.concatenate(new_w), new_gconsts=np.concatenate(new_gc),
               new_means_invvars=np.concatenate(new_miv), new_inv_vars=np.concatenate(new_iv), objf_change=objf, count=cnt)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "em_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; statuses", out["status_wide"], out["status_narrow"])


if __name__ == "__main__":
    main()
