"""Randomised parity sweep (not part of pytest): random model / utterance shapes and beams, HIP path vs the oracle:
alignments + status bit-exact, likelihood rel 2e-5, statistics (one-hot regime) rtol 2e-5, device M-step parameters
bit-exact vs the oracle on the device's statistics, gconsts <= 4 ulp.   python tests/manual/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
from helpers import build, exact_loglikes, oracle_graph, utt_feats
from oracle import oracle as orc
from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, MleDiagGmmOptions

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
t0 = time.time(); n = 0; nutt = 0; nfall = 0; nerr = 0
while time.time() - t0 < budget:
    P = int(rng.choice([3, 6, 12, 30, 60, 150]))
    G = int(rng.choice([1, 2, 3, 8, 16, 17, 32, 48, 64, 65, 100, 128]))
    D = int(rng.choice([1, 5, 13, 23, 39, 40, 41, 64, 80]))
    ragged = bool(rng.integers(2))
    lo = int(rng.choice([1, 2, 5, 20])); hi = lo + int(rng.choice([0, 2, 10, 30]))
    U = int(rng.choice([1, 3, 9, 20]))
    beam, retry = [(200.0, 0.0), (20.0, 0.0), (8.0, 40.0), (3.0, 10.0), (1.0, 2.0)][int(rng.integers(5))]
    seed = int(rng.integers(1 << 30))
    tag = f"P{P} G{G} D{D} ragged{int(ragged)} phones{lo}-{hi} U{U} beam{beam}/{retry} seed{seed}"
    m, gc, om, ut, cost = build(P, G, D, n_utt=U, seed=seed, ragged=ragged, min_phones=lo, max_phones=hi)
    if int(np.diff(ut.graphs["state_off"]).max()) > 1400:
        continue
    dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars, weights=m.weights)
    tm = DeviceTransitions(ctx, m.id2pdf); tm.set_trans_cost(cost)
    us = UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    # K1 against the fp64 bound (the tolerance of tests/test_gpu_parity.py)
    us.loglikes(dm)
    got_ll = us.download_loglikes()
    poff, pdfs = us.pdf_lists()
    mats = []
    for u in range(U):
        pl = pdfs[poff[u]: poff[u + 1]]
        exact, bound = exact_loglikes(m, gc, utt_feats(ut, u), pl)
        assert (np.abs(got_ll[u] - exact) <= 1e-5 + 1e-6 * bound).all(), (tag, u, "K1")
        mats.append(orc.loglikes_matrix(om, utt_feats(ut, u), pl))
    # K2 on IDENTICAL scores (the oracle's): alignment, status, words bit-exact whatever the beam does
    us.upload_loglikes(mats)
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1)
    oa = orc.OAccs(int(m.gauss_off[-1]), D, m.num_tids)
    ali_ok = np.zeros(ut.frame_off[-1], np.int32)
    for u in range(U):
        f = utt_feats(ut, u)
        T = int(ut.frame_off[u + 1] - ut.frame_off[u])
        want = orc.align_utterance_ll(oracle_graph(ut, u, cost), m.id2pdf, T, pdfs[poff[u]: poff[u + 1]], mats[u], acoustic_scale=0.1,
                                      beam=beam, retry_beam=retry)
        st = int(res["status"][u])
        assert (st & 3) == (want["status"] & 3), (tag, u, st, want["status"])
        nfall += (st & 8) != 0
        sl = slice(ut.frame_off[u], ut.frame_off[u + 1])
        if want["status"] & 1:
            nerr += 1
            assert (res["ali"][sl] == 0).all(), (tag, u)
            continue
        assert (res["ali"][sl] == want["ali"]).all(), (tag, u)
        assert abs(res["like"][u] - want["like"]) <= 1e-6 * abs(want["like"]) + 1e-4, (tag, u)
        ali_ok[sl] = want["ali"]
        orc.acc_stats_ali(om, m.id2pdf, f, want["ali"], oa)
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs)
    got = accs.download()
    assert (got["trans_acc"] == oa.trans_acc).all(), tag
    np.testing.assert_allclose(got["occ"], oa.occ, rtol=2e-4, atol=1e-5, err_msg=tag)
    np.testing.assert_allclose(got["mean_acc"], oa.mean_acc, rtol=2e-4, atol=2e-5 * max(1e-30, np.abs(oa.mean_acc).max()), err_msg=tag)
    np.testing.assert_allclose(got["var_acc"], oa.var_acc, rtol=2e-4, atol=2e-5 * max(1e-30, np.abs(oa.var_acc).max()), err_msg=tag)
    occ_min = float(rng.choice([0.5, 3.0, 10.0])); fl = int(rng.choice([7, 5, 4, 2, 3, 1]))
    r = dm.mle_update(accs, MleDiagGmmOptions(min_gaussian_occupancy=occ_min), fl)
    d = dm.download()
    for p in range(P):
        a, b = int(m.gauss_off[p]), int(m.gauss_off[p + 1])
        w = orc.mle_diag_gmm_update(m.weights[a:b], m.means_invvars[a:b], m.inv_vars[a:b], got["occ"][a:b], got["mean_acc"][a:b],
                                    got["var_acc"][a:b], acc_flags=0xF, flags=fl, min_gaussian_occupancy=occ_min)
        a2, b2 = int(d["gauss_off"][p]), int(d["gauss_off"][p + 1])
        assert b2 - a2 == len(w["weights"]), (tag, p, "removed")
        for k in ("weights", "inv_vars", "means_invvars"):
            assert np.array_equal(d[k][a2:b2], w[k]), (tag, p, k)
        # gconst = log w - D/2 log 2pi + sum_d (1/2 log iv - 1/2 miv^2 / iv), float accumulator: the logf difference shows up at
        # the ulp of the largest partial sum (with tiny D the terms can cancel to a much smaller result)
        ivf, mivf = w["inv_vars"].astype(np.float64), w["means_invvars"].astype(np.float64)
        big = np.abs(np.log(w["weights"].astype(np.float64))) + 0.5 * 1.8378770664093453 * D + (0.5 * np.abs(np.log(ivf)) + 0.5 * mivf * mivf / ivf).sum(1)
        dgc = np.abs(d["gconsts"][a2:b2] - w["gconsts"])
        if not (dgc <= 4 * np.spacing(big.astype(np.float32))).all():
            i = int((dgc / np.spacing(big.astype(np.float32))).argmax())
            raise AssertionError((tag, p, "gconsts", i, float(d["gconsts"][a2 + i]), float(w["gconsts"][i]), float(big[i]), float(w["weights"][i]),
                                  w["inv_vars"][i].tolist(), w["means_invvars"][i].tolist(), occ_min, fl))
    n += 1; nutt += U
    us.close(); accs.close(); tm.close(); dm.close()
print(f"fuzz ok: {n} configurations, {nutt} utterances ({nerr} oracle-failed, {nfall} through the fallback decoder) in {time.time() - t0:.0f}s")
