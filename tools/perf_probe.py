"""Per-kernel HIP-event timings of one EM pass on a mid-size synthetic set (dev aid)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import build
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, DeviceAccs
P, G, D, U = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (600, 64, 40, 2000)))
ctx = Context(0)
m, gc, om, ut, cost = build(P, G, D, n_utt=U, seed=3, min_phones=10, max_phones=40)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf); tm.set_trans_cost(cost)
us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
accs = DeviceAccs(ctx, dm, tm)
us.loglikes(dm)
res = us.align(tm, acoustic_scale=0.1)
print('frames', ut.frame_off[-1], 'status', np.unique(res['status'], return_counts=True), 'ali acc', (res['ali'] == ut.ref_ali).mean())
ctx.set_timing(True)
for i in range(3):
    accs.zero(); us.loglikes(dm); us.align(tm, acoustic_scale=0.1, download=False); us.acc_stats(dm, tm, accs)
tm_ = ctx.timings()
for n, ms in tm_[-5:]: print(n, round(ms, 3))
r = accs.download(); print('frames acc', r['total_frames'], 'avg ll', r['total_log_like'] / r['total_frames'])
