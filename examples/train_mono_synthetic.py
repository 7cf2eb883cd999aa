#!/usr/bin/env python3
"""Monophone EM training on a synthetic YES/NO task -- the flow of the reference's egs/yesno/train.py
(BASELINE.json configs[0]) through this package's drop-in names, end to end on one MI355X:

  generate_hmm_topo -> gmm_init_mono -> TrainingGraphCompiler.compile_graph_from_text ->
  equal_align -> gmm_acc_stats_ali_batch -> gmm_est -> {gmm_boost_silence, gmm_align_compiled_batch,
  gmm_acc_stats_ali_batch, gmm_est} x iterations

The reference's recipe needs the yesno audio, lhotse and kaldifst (none available offline); the
synthetic generator keeps its shape: words YES / NO with optional silence, 3-state phones, 5-state
silence, 23-dim features, beam 6 / retry 40, scales 0.1 / 1.0 / 0.1, the same realign schedule.

--resident runs the same schedule through khg.ResidentEm: features, graphs, alignments, accumulators and the
model stay in HBM across passes, the GMM update runs on the device (K4); only the passes that mix up go through
the host.

Usage: python examples/train_mono_synthetic.py [--utts 60] [--iters 20] [--resident]
"""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kaldi_hmm_gmm_amd as khg  # noqa: E402
from kaldi_hmm_gmm_amd.training_graph import TrainingGraphCompiler, equal_align, generate_hmm_topo  # noqa: E402

SIL, Y, N = 1, 2, 3            # phones (egs/yesno: SIL, Y, N)
YES, NO = 1, 2                 # words


def make_data(n_utt, dim, rng):
    """Each utterance: 3..8 random YES/NO words, silence around them with probability 0.5; every phone
    state emits from its own Gaussian for 3..8 frames."""
    nstate = {SIL: 5, Y: 3, N: 3}
    base = {SIL: 0, Y: 5, N: 8}
    means = (rng.standard_normal((11, dim)) * 2.5).astype(np.float32)
    utts = []
    for u in range(n_utt):
        words = [int(w) for w in rng.integers(1, 3, size=int(rng.integers(3, 9)))]
        phones = [SIL] if rng.random() < 0.5 else []
        for w in words:
            phones.append(Y if w == YES else N)
            if rng.random() < 0.5:
                phones.append(SIL)
        x = []
        for ph in phones:
            for s in range(nstate[ph]):
                d = int(rng.integers(3, 9))
                x.append(means[base[ph] + s] + rng.standard_normal((d, dim)).astype(np.float32))
        utts.append((f"utt{u:04d}", words, np.concatenate(x).astype(np.float32)))
    return utts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=60)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dim", type=int, default=23)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--resident", action="store_true", help="keep everything in HBM across passes (khg.ResidentEm)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    utts = make_data(args.utts, args.dim, rng)
    names = [u[0] for u in utts]
    feats = [u[2] for u in utts]

    topo = generate_hmm_topo(non_sil_phones=[Y, N], sil_phone=SIL)
    transition_model, tree, am = khg.gmm_init_mono(topo, np.concatenate(feats[:10]))     # egs/yesno/train.py:49-50
    lexicon = {YES: [(1.0, [Y])], NO: [(1.0, [N])]}
    gc = TrainingGraphCompiler(transition_model, tree, lexicon, sil_phone=SIL, sil_prob=0.5)
    train_graphs = gc.compile_graphs_from_text([u[1] for u in utts])                   # :70-83

    ali = []
    for g, x in zip(train_graphs, feats):                                               # :86-108
        ok, a = equal_align(g, x.shape[0], rand_seed=3, num_retries=10)
        if not ok:
            raise SystemExit("equal_align failed")
        ali.append(a)

    if args.resident:
        return train_resident(args, utts, names, feats, transition_model, tree, am, train_graphs, ali)

    def accumulate():
        accs = khg.AccumAmDiagGmm()
        accs.init(am, khg.GmmUpdateFlags.kGmmAll)
        ll, tacc = khg.gmm_acc_stats_ali_batch(am, accs, transition_model, feats, ali)
        return accs, tacc, ll

    num_gauss = am.num_pdfs                                                             # one Gaussian per pdf
    max_gauss = 4 * am.num_pdfs
    max_iter_inc = max(1, args.iters * 3 // 8)
    inc_gauss = (max_gauss - num_gauss) // max_iter_inc
    tcfg = khg.MleTransitionUpdateConfig()
    accs, tacc, ll = accumulate()
    opts = khg.MleDiagGmmOptions()
    opts.min_gaussian_occupancy = 3
    randn = seeded_randn(args.seed + 1)
    khg.gmm_est(am, accs, transition_model, tacc, tcfg, opts, mixup=num_gauss, update_flags="mvwt", verbose=False, randn=randn)   # :131-150
    realign = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 38, 40, 42, 44, 46, 49, 52, 55, 58, 60, 65, 70, 75, 78, 79}   # egs/yesno/train.py:153
    cfg = khg.AlignConfig(beam=6.0, retry_beam=40.0, careful=False)
    for i in range(args.iters):
        if i in realign:
            am_b = khg.gmm_boost_silence(am, transition_model, [SIL], boost=1.0)    # a boosted COPY aligns; `am` is what gets updated
            r = khg.gmm_align_compiled_batch(am_b, transition_model, names, train_graphs, feats, cfg, acoustic_scale=0.1,
                                             transition_scale=1.0, self_loop_scale=0.1)
            ali = [a if a else old for a, old in zip(r["alignment"], ali)]
            print(f"pass {i}: aligned {r['num_done']} utterances, {r['num_error']} errors, {r['num_retried']} retried, "
                  f"avg like/frame {r['tot_like'] / max(r['frame_count'], 1):.4f}")
        accs, tacc, ll = accumulate()
        info = khg.gmm_est(am, accs, transition_model, tacc, tcfg, khg.MleDiagGmmOptions(), mixup=num_gauss, perturb_factor=0.01,
                           power=0.2, min_count=20.0, update_flags="mvwt", verbose=False, randn=randn)
        print(f"pass {i}: avg log-like per frame {info['avg_like']:.4f} over {info['frames']:.0f} frames, "
              f"{am.num_gauss} Gaussians")
        if i < max_iter_inc:
            num_gauss += inc_gauss
    # word recovery: the aligned phone sequence must spell the transcript
    r = khg.gmm_align_compiled_batch(am, transition_model, names, train_graphs, feats, cfg, acoustic_scale=0.1,
                                     transition_scale=1.0, self_loop_scale=0.1)
    ok = sum(1 for w, u in zip(r["words"], utts) if w == u[1])
    print(f"final: {ok}/{len(utts)} utterances aligned to their transcript; {r['num_error']} errors")
    if args.out:
        with open(args.out, "wb") as fh:                                                # :224-229 (torch.save of pickles)
            pickle.dump({"acoustic_model": am, "transition_model": transition_model, "tree": tree}, fh)
    return 0 if ok == len(utts) and r["num_error"] == 0 else 1


def seeded_randn(seed):
    """DiagGmm::Split draws its perturbations from the global RNG in the reference; a seeded stream makes a run reproducible."""
    rng = np.random.default_rng(seed)
    return lambda d: rng.standard_normal(d).astype(np.float32)


def train_resident(args, utts, names, feats, transition_model, tree, am, train_graphs, ali, randn=None, log=print):
    """The schedule of main() with the shard resident on the GPU."""
    randn = randn or seeded_randn(getattr(args, "seed", 3) + 1)
    em = khg.ResidentEm(am, transition_model, train_graphs, feats, acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
    em.set_alignments(ali)
    num_gauss = am.num_pdfs
    max_gauss = 4 * am.num_pdfs
    max_iter_inc = max(1, args.iters * 3 // 8)
    inc_gauss = (max_gauss - num_gauss) // max_iter_inc
    tcfg = khg.MleTransitionUpdateConfig()
    opts = khg.MleDiagGmmOptions()
    opts.min_gaussian_occupancy = 3
    em.accumulate()
    em.update(tcfg, opts, mixup=num_gauss, update_flags="mvwt", randn=randn)
    realign = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 38, 40, 42, 44, 46, 49, 52, 55, 58, 60, 65, 70, 75, 78, 79}   # egs/yesno/train.py:153
    cfg = khg.AlignConfig(beam=6.0, retry_beam=40.0, careful=False)
    for i in range(args.iters):
        if i in realign:
            em.boost_silence([SIL], boost=1.0)
            r = em.align(cfg)
            log(f"pass {i}: aligned {r['num_done']} utterances, {r['num_error']} errors, {r['num_retried']} retried, "
                f"avg like/frame {r['tot_like'] / max(r['frame_count'], 1):.4f}")
        em.accumulate()
        info = em.update(tcfg, khg.MleDiagGmmOptions(), mixup=num_gauss, perturb_factor=0.01, power=0.2, min_count=20.0,
                         update_flags="mvwt", randn=randn)
        log(f"pass {i}: avg log-like per frame {info['avg_like']:.4f} over {info['frames']:.0f} frames, "
            f"{em.num_gauss} Gaussians")
        if i < max_iter_inc:
            num_gauss += inc_gauss
    em.sync_host()
    em.close()
    r = khg.gmm_align_compiled_batch(am, transition_model, names, train_graphs, feats, cfg, acoustic_scale=0.1,
                                     transition_scale=1.0, self_loop_scale=0.1)
    ok = sum(1 for w, u in zip(r["words"], utts) if w == u[1])
    log(f"final: {ok}/{len(utts)} utterances aligned to their transcript; {r['num_error']} errors")
    if args.out:
        with open(args.out, "wb") as fh:
            pickle.dump({"acoustic_model": am, "transition_model": transition_model, "tree": tree}, fh)
    return 0 if ok == len(utts) and r["num_error"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
