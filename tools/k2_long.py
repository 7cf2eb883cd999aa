"""K1 / K2 / K3 against utterance length: the benchmark's model (5000 x 64 x 40), ~6 M frames per set, beam 200 (exact DP, no fallback).
usage: k2_long.py [frames_in_millions]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth
from oracle import oracle as orc   # gconsts only (test rig)
P, G, D = 5000, 64, 40
MF = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
m = synth.make_model(P, G, D, seed=20230417)
gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
ctx = Context(0)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf)
il = np.arange(m.num_tids + 1, dtype=np.int32)
cost = orc.add_transition_probs(il, np.zeros(m.num_tids + 1, np.float32), m.log_probs, m.non_self_loop_log_probs, m.id2state, m.is_self_loop, 1.0, 0.1)
tm.set_trans_cost(cost)
import os
RANGES = eval(os.environ.get('K2_LONG_RANGES', '((10, 40), (40, 100), (100, 200), (200, 330), (340, 600))'))
for lo, hi in RANGES:
    U = int(MF * 1e6 / (12.0 * (lo + hi) / 2))
    ut = synth.make_utts(m, U, seed=5, min_phones=lo, max_phones=hi, feats=False)
    feats = synth.sample_feats_torch(m, ut.frame_pdf, 7, torch.device("cuda", 0))
    us = UtteranceSet(ctx, tm, ut.frame_off, (feats.data_ptr(), feats), dim=D, graphs=ut.graphs)
    accs = DeviceAccs(ctx, dm, tm)
    for rep in range(2):
        ctx.set_timing(rep == 1)
        us.loglikes(dm, band=True)
        res = us.align(tm, beam=200.0, acoustic_scale=0.1, download=False)
        us.acc_stats(dm, tm, accs)
        ctx.sync()
    km = dict(ctx.timings()); ctx.set_timing(False)
    N = int(ut.frame_off[-1])
    S = np.diff(ut.graphs["state_off"])
    print(f"{lo}..{hi} phones: {U} utts, {N/1e6:.2f} M frames, states {S.min()}..{S.max()} | " + ", ".join(f"{k} {v:.2f}" for k, v in km.items()) +
          f" | K2 {km.get('k2_viterbi_dp', 0) / N * 1e6:.3f} ns/frame, K1 {km.get('k1_loglikes', 0) / N * 1e6:.3f} ns/frame")
    us.close(); accs.close(); del feats
