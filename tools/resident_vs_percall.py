"""Debug: per-call vs resident EM flows, pass-by-pass differences (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import kaldi_hmm_gmm_amd as khg
from kaldi_hmm_gmm_amd.training_graph import TrainingGraphCompiler, equal_align, generate_hmm_topo
import train_mono_synthetic as ex

rng = np.random.default_rng(11)
utts = ex.make_data(20, 13, rng)
names, feats = [u[0] for u in utts], [u[2] for u in utts]
topo = generate_hmm_topo(non_sil_phones=[ex.Y, ex.N], sil_phone=ex.SIL)

def fresh():
    tm, tree, am = khg.gmm_init_mono(topo, np.concatenate(feats[:10]))
    comp = TrainingGraphCompiler(tm, tree, {ex.YES: [(1.0, [ex.Y])], ex.NO: [(1.0, [ex.N])]}, sil_phone=ex.SIL, sil_prob=0.5)
    graphs = comp.compile_graphs_from_text([u[1] for u in utts])
    ali = []
    for g, x in zip(graphs, feats):
        ok, a = equal_align(g, x.shape[0], rand_seed=3, num_retries=10)
        ali.append(a)
    return tm, am, graphs, ali

def randn(seed):
    r = np.random.default_rng(seed)
    return lambda d: r.standard_normal(d).astype(np.float32)

cfg = khg.AlignConfig(beam=6.0, retry_beam=40.0, careful=False)
tcfg = khg.MleTransitionUpdateConfig()
opts = khg.MleDiagGmmOptions(min_gaussian_occupancy=3)
mix = [11, 16, 22, 22, 22]

def flow_a():
    tm, am, graphs, ali = fresh(); rn = randn(5); out = []
    for it, target in enumerate(mix):
        if it > 0:
            am_b = khg.gmm_boost_silence(am, tm, [ex.SIL], boost=1.25)      # a boosted COPY aligns (the reference's semantics)
            r = khg.gmm_align_compiled_batch(am_b, tm, names, graphs, feats, cfg, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
            ali = r["alignment"]
        accs = khg.AccumAmDiagGmm(); accs.init(am, khg.GmmUpdateFlags.kGmmAll)
        ll, tacc = khg.gmm_acc_stats_ali_batch(am, accs, tm, feats, ali)
        khg.gmm_est(am, accs, tm, tacc, tcfg, opts, mixup=target, update_flags="mvwt", verbose=False, randn=rn)
        out.append((np.concatenate([np.asarray(a) for a in ali]), [x.copy() for x in am.flat()]))
    return out

def flow_b():
    tm, am, graphs, ali = fresh(); rn = randn(5); out = []
    em = khg.ResidentEm(am, tm, graphs, feats, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
    em.set_alignments(ali)
    for it, target in enumerate(mix):
        if it > 0:
            em.boost_silence([ex.SIL], boost=1.25)
            em.align(cfg)
        em.accumulate()
        em.update(tcfg, opts, mixup=target, update_flags="mvwt", randn=rn)
        em.sync_host()
        out.append((em.us.download_ali(), [x.copy() for x in am.flat()]))
    return out

def cmp(x, y, name):
    for it, ((a1, m1), (a2, m2)) in enumerate(zip(x, y)):
        same_shape = all(u.shape == v.shape for u, v in zip(m1, m2))
        d = [float(np.abs(u.astype(np.float64) - v).max()) if same_shape else -1 for u, v in zip(m1, m2)]
        print(name, "pass", it, "ali mismatches", int((a1 != a2).sum()), "of", a1.size, "gauss", m1[0][-1], m2[0][-1],
              "max|d| go,gc,w,miv,iv:", ["%.3g" % v for v in d])

A1, A2, B1, B2 = flow_a(), flow_a(), flow_b(), flow_b()
cmp(A1, A2, "A-A"); cmp(B1, B2, "B-B"); cmp(A1, B1, "A-B")

# sensitivity: flow A up to the pass-3 alignment, then align with gconsts nudged by +-1 ulp on random Gaussians
tm, am, graphs, ali = fresh(); rn = randn(5)
for it, target in enumerate(mix[:4]):
    if it > 0:
        am_b = khg.gmm_boost_silence(am, tm, [ex.SIL], boost=1.25)
        if it == 3:
            break
        r = khg.gmm_align_compiled_batch(am_b, tm, names, graphs, feats, cfg, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
        ali = r["alignment"]
    accs = khg.AccumAmDiagGmm(); accs.init(am, khg.GmmUpdateFlags.kGmmAll)
    ll, tacc = khg.gmm_acc_stats_ali_batch(am, accs, tm, feats, ali)
    khg.gmm_est(am, accs, tm, tacc, tcfg, opts, mixup=target, update_flags="mvwt", verbose=False, randn=rn)
base = khg.gmm_align_compiled_batch(am, tm, names, graphs, feats, cfg, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
b = np.concatenate([np.asarray(a) for a in base["alignment"]])
go, gc, w, miv, iv = am.flat()
pr = np.random.default_rng(0)
for trial in range(6):
    g2 = gc.copy()
    sel = pr.random(g2.shape[0]) < 0.3
    g2[sel] = np.nextafter(g2[sel], np.where(pr.random(sel.sum()) < 0.5, np.inf, -np.inf).astype(np.float32))
    am.set_flat(go, w, g2, miv, iv)
    r = khg.gmm_align_compiled_batch(am, tm, names, graphs, feats, cfg, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
    c = np.concatenate([np.asarray(a) for a in r["alignment"]])
    print("1-ulp gconst nudge trial", trial, "nudged", int(sel.sum()), "of", sel.size, "-> ali mismatches", int((b != c).sum()))
