import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from helpers import build, oracle_replay
from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet
m, gc, om, ut, cost = build(200, 64, 40, n_utt=60, seed=5)
keep = oracle_replay(m, gc, ut, cost, 60)
ctx = Context(0)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf)
sub = UtteranceSet(ctx, None, ut.frame_off.astype(np.int64), ut.feats)
sub.upload_ali(np.ascontiguousarray(keep["ali"], np.int32))
accs = DeviceAccs(ctx, dm, tm)
sub.acc_stats(dm, tm, accs)
got = accs.download()
oa = keep["accs"]
d = np.abs(got["occ"]-oa.occ)
print("occ max abs diff", d.max(), "max rel", (d/(np.abs(oa.occ)+1e-30)).max(), "n exactly equal", (d==0).mean(), oa.occ[:5], got["occ"][:5])
d = np.abs(got["mean_acc"]-oa.mean_acc); print("mean max abs", d.max(), (d==0).mean())
