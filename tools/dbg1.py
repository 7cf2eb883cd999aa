import sys, time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from helpers import build
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet
ctx = Context(0)
m, gc, om, ut, cost = build(300, 8, 40, n_utt=200, seed=3, min_phones=10, max_phones=40)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf); tm.set_trans_cost(cost)
us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
us.loglikes(dm)
res = us.align(tm, acoustic_scale=0.1)
st = res['status']
print('status histogram', np.unique(st, return_counts=True))
print('ali acc', (res['ali'] == ut.ref_ali).mean())
ctx.set_timing(True)
for i in range(3):
    us.loglikes(dm); us.align(tm, acoustic_scale=0.1, download=False)
for n, ms in ctx.timings(): print(n, round(ms, 3))
