"""One-off larger end-to-end check (not part of pytest): K1 -> K2 -> K3 through the C-ABI vs the oracle pipeline on a few
hundred bench-like utterances (long, ~75 pdfs each, 64 Gaussians)."""
import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from helpers import build, oracle_graph, utt_feats
from oracle import oracle as orc
from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet

U = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m, gc, om, ut, cost = build(600, 64, 40, n_utt=U, seed=91, min_phones=10, max_phones=40)
ctx = Context(0)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf); tm.set_trans_cost(cost)
us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
for beam, retry in ((200.0, 0.0), (6.0, 40.0)):
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1)
    accs = DeviceAccs(ctx, dm, tm); us.acc_stats(dm, tm, accs); got = accs.download()
    oa = orc.OAccs(int(m.gauss_off[-1]), m.dim, m.num_tids)
    t0 = time.time(); bad = 0; like_err = 0.0
    for u in range(U):
        f = utt_feats(ut, u)
        want = orc.align_utterance(oracle_graph(ut, u, cost), om, m.id2pdf, f, acoustic_scale=0.1, beam=beam, retry_beam=retry)
        a = res["ali"][ut.frame_off[u]: ut.frame_off[u + 1]]
        if (int(res["status"][u]) & 3) != (want["status"] & 3) or (not (want["status"] & 1) and not (a == want["ali"]).all()):
            bad += 1
        if not (want["status"] & 1):
            like_err = max(like_err, abs(float(res["like"][u]) - want["like"]) / max(1.0, abs(want["like"])))
            orc.acc_stats_ali(om, m.id2pdf, f, want["ali"], oa)
    occ_err = np.abs(got["occ"] - oa.occ).max() / max(1.0, np.abs(oa.occ).max())
    mean_err = np.abs(got["mean_acc"] - oa.mean_acc).max() / max(1.0, np.abs(oa.mean_acc).max())
    var_err = np.abs(got["var_acc"] - oa.var_acc).max() / max(1.0, np.abs(oa.var_acc).max())
    print("   sums: occ gpu/oracle", float(got["occ"].sum()), float(np.asarray(oa.occ).sum()), " mean_acc", float(np.abs(got["mean_acc"]).sum()), float(np.abs(oa.mean_acc).sum()))
    print(f"beam {beam}/{retry}: {U} utts, {int(ut.frame_off[-1])} frames: alignment mismatches {bad}, max rel like err {like_err:.2e}, "
          f"trans_acc equal {bool((got['trans_acc'] == oa.trans_acc).all())}, occ/mean/var max err (rel to max) {occ_err:.1e} {mean_err:.1e} {var_err:.1e}, "
          f"fallback {int(((res['status'] & 8) != 0).sum())}, oracle {time.time() - t0:.1f}s")
