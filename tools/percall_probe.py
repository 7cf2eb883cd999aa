#!/usr/bin/env python3
"""Times the reference's own call pattern (egs/yesno/train.py:170-202): gmm_align_compiled + gmm_acc_stats_ali, one call per
utterance, host numpy features and StdVectorFst graphs, at a synthetic model size.  usage: percall_probe.py [config] [n_utt]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kaldi_hmm_gmm_amd as khg
from kaldi_hmm_gmm_amd import synth


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "tri5000x64"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    P, G, D = synth.CONFIGS[cfg]
    m = synth.make_model(P, G, D, seed=20230418)
    ut = synth.make_utts(m, n, seed=5)
    t0 = time.perf_counter()
    am, tm = synth.host_objects(m)
    print(f"host objects: {time.perf_counter() - t0:.2f} s", flush=True)
    cfg_a = khg.AlignConfig(beam=200.0, retry_beam=0.0, careful=False)
    fsts = [synth.utt_fst(ut.graphs, u) for u in range(n)]
    feats = [np.ascontiguousarray(ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]]) for u in range(n)]
    accs = khg.AccumAmDiagGmm()
    accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    tacc = None
    ta, tb, frames = [], [], 0
    for u in range(n):
        t0 = time.perf_counter()
        r = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt=str(u), fst=fsts[u].copy(), feats=feats[u], align_config=cfg_a,
                                   acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
        t1 = time.perf_counter()
        ll, tacc = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=feats[u], ali=r["alignment"], transition_accs=tacc)
        t2 = time.perf_counter()
        ta.append(t1 - t0); tb.append(t2 - t1); frames += feats[u].shape[0]
        if u < 3 or u == n - 1:
            print(f"utt {u}: T {feats[u].shape[0]} align {1e3 * (t1 - t0):.2f} ms, acc {1e3 * (t2 - t1):.2f} ms, done {r['num_done']}", flush=True)
    k = min(5, n // 2)
    sa, sb = sum(ta[k:]), sum(tb[k:])
    fr = sum(f.shape[0] for f in feats[k:])
    t0 = time.perf_counter()
    tot = accs.tot_count
    t1 = time.perf_counter()
    print(f"{cfg}: {n - k} utts after warm-up: align {1e3 * sa / (n - k):.3f} ms/utt, acc {1e3 * sb / (n - k):.3f} ms/utt, "
          f"{fr / (sa + sb):.0f} frames/s; tot_count {tot} (first host read {1e3 * (t1 - t0):.1f} ms)")
    # the reference's own loop body (scripts/gmm_acc_stats_ali.py:46-56): one accumulate_for_gmm per FRAME -- launch-bound by construction
    acc2 = khg.AccumAmDiagGmm()
    acc2.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    ali = np.asarray(r["alignment"], np.int32)
    pdfs = [tm.transition_id_to_pdf(int(t)) for t in ali]
    f = feats[n - 1]
    for t in range(8):
        acc2.accumulate_for_gmm(model=am, data=f[t], gmm_index=pdfs[t], weight=1.0)
    t0 = time.perf_counter()
    for t in range(f.shape[0]):
        acc2.accumulate_for_gmm(model=am, data=f[t], gmm_index=pdfs[t], weight=1.0)
    dt = time.perf_counter() - t0
    print(f"per-frame accumulate_for_gmm: {1e3 * dt / f.shape[0]:.3f} ms per frame = {f.shape[0] / dt:.0f} frames/s")


if __name__ == "__main__":
    main()
