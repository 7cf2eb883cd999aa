"""BASELINE.json configs[4] at its REAL model shape -- 10 000 pdfs x 128 Gaussians x 80 dims (1.28 M Gaussians, 824 MB tile
image, 1.65 GB fp64 accumulator block) -- on ~1000 bench-like utterances: the oracle replays the first 60 of them (identical
alignments, statistics to 2e-5), the whole set goes through size-independent properties, plus the "fp32 stats vs fp64" accumulator tolerance report that config names:

  K1  pdf-major form with 8 register-resident row blocks (D > 40: KQ = 20) vs the utterance-major form: <= 2 float ulps
  K2  accepting path, returned likelihood == host replay in the token arithmetic, never worse than the generating path
  K3  block form (128 Gaussians, D = 80): sum(occ) = frames, transition counts = histogram of the alignment,
      sum_g mean_acc = sum_t x_t, sum_g var_acc = sum_t x_t^2
  K4  fixed point: statistics that are the model's own moments give the model back
  C1  eight shards' blocks summed in fp32 (the fp32-wire all-reduce) vs in fp64: post-M-step parameter error, reported
"""
import json
import os

import numpy as np
import pytest

from helpers import token_path_cost
from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, MleDiagGmmOptions, UtteranceSet, synth
from kaldi_hmm_gmm_amd.dist import shard_utterances, take_utterances
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, G, D, U = 10000, 128, 80, 1000


@pytest.fixture(scope="module")
def stress(ctx):
    m = synth.make_model(P, G, D, seed=20230419)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    ut = synth.make_utts(m, U, seed=6)
    il = np.arange(m.num_tids + 1, dtype=np.int32)
    cost = orc.add_transition_probs(il, np.zeros(m.num_tids + 1, np.float32), m.log_probs, m.non_self_loop_log_probs,
                                    m.id2state, m.is_self_loop, 1.0, 0.1)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    tm = DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    yield m, gc, ut, cost, dm, tm, us
    for o in (us, tm, dm):
        o.close()


def test_config5_em_pass_properties(ctx, stress, opt):
    m, gc, ut, cost, dm, tm, us = stress
    N = int(ut.frame_off[-1])
    assert N > 250000
    # ---- K1: two fp32-MFMA tilings, one answer; the f16x2s form (the default) checked against fp64 below ----
    opt.k1("pdf")
    us.loglikes(dm)
    ll = us.download_loglikes()
    opt.k1("utt")
    us.loglikes(dm)
    ll_utt = us.download_loglikes()
    opt.k1("auto")
    us.loglikes(dm)
    ll_b = us.download_loglikes()
    assert max(float(np.abs(x - y).max()) for x, y in zip(ll, ll_b)) < 5e-3 and all(np.isfinite(x).all() for x in ll_b)
    worst = max(float(np.max(np.abs(x - y) / np.spacing(np.abs(x)))) for x, y in zip(ll, ll_utt))
    same = sum(int((x == y).sum()) for x, y in zip(ll, ll_utt)) / sum(x.size for x in ll)
    assert worst <= 2.0 and same > 0.5, (worst, same)
    assert all(np.isfinite(x).all() for x in ll)
    # the default K1 and the fp32-MFMA form against an fp64 evaluation (the tolerance of tests/test_gpu_parity.py: 1e-5 + 1e-6 B):
    # every cell of 50 utterances spread over the set
    from helpers import exact_loglikes
    poff, pdfs = us.pdf_lists()
    for u in np.linspace(0, U - 1, 50).astype(int):
        pl = pdfs[poff[u]: poff[u + 1]]
        exact, bound = exact_loglikes(m, gc, ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]], pl)
        tol = 1e-5 + 1e-6 * bound
        assert (np.abs(ll[u] - exact) <= tol).all() and (np.abs(ll_b[u] - exact) <= tol).all(), u

    # ---- K2 ----
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=200.0, acoustic_scale=0.1)
    assert not np.any(res["status"] & 1)
    rng = np.random.default_rng(0)
    for u in rng.choice(U, size=60, replace=False):
        sl = slice(ut.frame_off[u], ut.frame_off[u + 1])
        pl = pdfs[poff[u]: poff[u + 1]]
        ok, c = token_path_cost(ut.graphs, u, res["ali"][sl], ll_b[u], pl, cost, m.id2pdf, 0.1)
        assert ok, f"utterance {u}: not an accepting path"
        assert -c / 0.1 == pytest.approx(float(res["like"][u]), rel=2e-6)
        ok_ref, c_ref = token_path_cost(ut.graphs, u, ut.ref_ali[sl], ll_b[u], pl, cost, m.id2pdf, 0.1)
        assert ok_ref and c <= c_ref + 1e-9

    # the oracle's FasterDecoder + acc-stats on the first 60 utterances of this set (4 threads: every thread owns a 1.65 GB
    # accumulator set at this shape): identical alignments, K3's statistics of those utterances to 2e-5
    from helpers import assert_matches_oracle_replay, oracle_replay
    keep = oracle_replay(m, gc, ut, cost, 60, acoustic_scale=0.1, beam=200.0, threads=4)
    assert_matches_oracle_replay(ctx, dm, tm, ut, res, keep, D)
    del keep

    # ---- K3 (block form: 128 Gaussians per pdf, D = 80) ----
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs)
    got = accs.download()
    accs.close()
    assert got["total_frames"] == N
    assert got["occ"].sum() == pytest.approx(N, rel=1e-6)
    assert np.array_equal(got["trans_acc"], np.bincount(res["ali"], minlength=m.num_tids + 1).astype(np.float64))
    x = ut.feats.astype(np.float64)
    np.testing.assert_allclose(got["mean_acc"].sum(0), x.sum(0), rtol=1e-5, atol=1e-6 * np.abs(x).sum(0).max())
    np.testing.assert_allclose(got["var_acc"].sum(0), (x * x).sum(0), rtol=1e-5)
    per_pdf = np.add.reduceat(got["occ"], m.gauss_off[:-1].astype(np.int64))
    np.testing.assert_allclose(per_pdf, np.bincount(m.id2pdf[res["ali"]], minlength=P), rtol=1e-5, atol=1e-4)


def test_config5_device_m_step_fixed_point(ctx, stress):
    m, gc, ut, cost, dm0, tm, us = stress
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    accs = DeviceAccs(ctx, dm, tm)
    mu = m.means_invvars.astype(np.float64) / m.inv_vars.astype(np.float64)
    var = 1.0 / m.inv_vars.astype(np.float64)
    occ = 12000.0 * m.weights.astype(np.float64)
    buf = np.zeros(accs.size, np.float64)
    sumG = int(m.gauss_off[-1])
    buf[:sumG] = occ
    buf[sumG: sumG + sumG * D] = (occ[:, None] * mu).ravel()
    buf[sumG + sumG * D: sumG + 2 * sumG * D] = (occ[:, None] * (var + mu * mu)).ravel()
    accs.upload(buf)
    del buf
    r = dm.mle_update(accs, MleDiagGmmOptions(), 0x7)
    assert r["removed"] == 0 and r["floored_elements"] == 0
    d = dm.download()
    np.testing.assert_allclose(d["weights"], m.weights, rtol=2e-6)
    np.testing.assert_allclose(d["inv_vars"], m.inv_vars, rtol=1e-4)
    np.testing.assert_allclose(d["means_invvars"], m.means_invvars, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d["gconsts"], gc, rtol=1e-5, atol=2e-4)
    accs.close()
    dm.close()


def test_config5_fp32_accumulator_exchange_vs_fp64(ctx, stress):
    """configs[4]: "fp32 stats vs CPU tolerance check".  The set is dealt to 8 shards as for 8 GPUs; each shard's block
    comes from K3 on the generating alignment.  (a) fp64 exchange: the 8 blocks added in fp64 (khg_accs_allreduce);
    (b) fp32 wire: every block rounded to fp32 and the 8 added in fp32, widened back (khg_accs_allreduce_f32).  Both
    go through K4; the parameter differences are reported and bounded."""
    import torch

    m, gc, ut, cost, dm0, tm, us_all = stress
    shards = shard_utterances(np.diff(ut.frame_off), 8)
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    accs = DeviceAccs(ctx, dm, tm)
    view = accs.as_torch()
    sum64 = torch.zeros_like(view)
    sum32 = torch.zeros(view.shape, dtype=torch.float32, device=view.device)
    for idx in shards:
        fo, g, fr = take_utterances(ut.frame_off, None, idx)
        us = UtteranceSet(ctx, None, fo, ut.feats[fr])
        us.upload_ali(ut.ref_ali[fr])
        accs.zero()
        us.acc_stats(dm, tm, accs)
        ctx.sync()
        torch.cuda.synchronize()
        sum64 += view
        accs.allreduce(None, wire_fp32=True)            # the wire image of this rank's block
        ctx.sync()
        sum32 += view.float()                           # fp32 ring sum
        torch.cuda.synchronize()
        us.close()
    # ~0.23 frames per Gaussian on 1000 utterances: keep what has data; 2.5 (not an integer: the one-hot posteriors make the
    # occupancies near-integers, and a threshold ON an integer would turn last-bit differences into different removals)
    opts = MleDiagGmmOptions(min_gaussian_occupancy=2.5)
    out = {}
    for name, blk in (("fp64", sum64), ("fp32", sum32.double())):
        view.copy_(blk)
        torch.cuda.synchronize()
        dmx = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
        r = dmx.mle_update(accs, opts, 0x7)
        out[name] = (r, dmx.download())
        dmx.close()
        accs.relayout(dm)
    (r64, d64), (r32, d32) = out["fp64"], out["fp32"]
    assert np.array_equal(d64["gauss_off"], d32["gauss_off"]) and r64["removed"] == r32["removed"]
    mu64 = d64["means_invvars"].astype(np.float64) / d64["inv_vars"]
    mu32 = d32["means_invvars"].astype(np.float64) / d32["inv_vars"]
    var64, var32 = 1.0 / d64["inv_vars"].astype(np.float64), 1.0 / d32["inv_vars"].astype(np.float64)
    rep = {"shape": {"pdfs": P, "gauss": G, "dim": D}, "utterances": U, "frames": int(ut.frame_off[-1]), "shards": 8,
           "gaussians_after": int(d64["gauss_off"][-1]), "removed": r64["removed"],
           "weights_max_rel": float(np.max(np.abs(d32["weights"] - d64["weights"]) / d64["weights"])),
           "means_max_abs": float(np.max(np.abs(mu32 - mu64))), "means_max_abs_over_sigma": float(np.max(np.abs(mu32 - mu64) / np.sqrt(var64))),
           "vars_max_abs": float(np.max(np.abs(var32 - var64))), "vars_max_rel": float(np.max(np.abs(var32 - var64) / var64)),
           "second_moment_scale": float(np.max(var64 + mu64 * mu64)),
           "gconsts_max_abs": float(np.max(np.abs(d32["gconsts"] - d64["gconsts"]))),
           "objf_change_fp64": r64["objf_change"], "objf_change_fp32": r32["objf_change"],
           "block_doubles": int(accs.size), "wire_bytes_fp64": int(accs.size) * 8, "wire_bytes_fp32": int(accs.size) * 4}
    for d_ in (os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d_):
            with open(os.path.join(d_, "r6_fp32_accs_report.json"), "w") as fh:
                json.dump(rep, fh, indent=1)
    # fp32 partial sums: 2^-24 relative per rounding, a few roundings per cell.  Means ~ +-10: 1e-5 absolute; the variance
    # is the difference E[x^2] - mu^2 of quantities up to ~100x larger than itself, so it is bounded in absolute terms
    # (its relative error, largest where few frames gave a tiny variance, is in the report)
    assert rep["weights_max_rel"] < 2e-6 and rep["means_max_abs"] < 1e-5 and rep["vars_max_abs"] < 1e-4, rep
    accs.close()
    dm.close()
