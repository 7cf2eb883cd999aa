import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import kaldi_hmm_gmm_amd as khg
from kaldi_hmm_gmm_amd import synth, _gpu
ctx = _gpu.default_context()
P, G, D = synth.CONFIGS["tri5000x64"]
m = synth.make_model(P, G, D, seed=20230418)
n = 40
ut = synth.make_utts(m, n, seed=5)
am, tm = synth.host_objects(m)
cfg_a = khg.AlignConfig(beam=200.0, retry_beam=0.0, careful=False)
fsts = [synth.utt_fst(ut.graphs, u) for u in range(n)]
feats = [np.ascontiguousarray(ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]]) for u in range(n)]
accs = khg.AccumAmDiagGmm(); accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
def one(u):
    r = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt=str(u), fst=fsts[u].copy(), feats=feats[u], align_config=cfg_a, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
    khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=feats[u], ali=r["alignment"], transition_accs=None)
    return r
for u in range(8): one(u)
ctx.set_timing(True)
ctx.timings()
t0 = time.perf_counter()
for u in range(8, n): one(u)
wall = time.perf_counter() - t0
tm_ = ctx.timings()
ctx.set_timing(False)
agg = {}
for k, v in tm_: agg[k] = agg.get(k, 0.0) + v
print("wall per utt (timing on): %.3f ms" % (1e3 * wall / (n - 8)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]): print("  %-28s %.4f ms per utt" % (k, v / (n - 8)))
t0 = time.perf_counter()
for u in range(8, n): one(u)
print("wall per utt (timing off): %.3f ms" % (1e3 * (time.perf_counter() - t0) / (n - 8)))
# host-only pieces
t0 = time.perf_counter()
for u in range(8, n): fsts[u].copy()
print("fst.copy per utt: %.3f ms" % (1e3 * (time.perf_counter() - t0) / (n - 8)))
