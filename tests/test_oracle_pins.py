"""Pin the CPU oracle against every known-answer / formula test the reference holds for this path
(SURVEY.md 8c).  The reference's tests re-evaluate closed-form formulas in torch on seeded random
inputs; the same formulas are re-evaluated here in numpy float64 and the oracle must meet the
reference's own tolerances (file:line cited per test)."""
import math

import os

import numpy as np
import pytest

from oracle import oracle as orc


def _rand_gmm(rng, nmix, dim, var_lo=0.05):
    w = rng.random(nmix).astype(np.float32)
    w /= w.sum()
    mean = rng.random((nmix, dim)).astype(np.float32)
    var = (rng.random((nmix, dim)) * (1 - var_lo) + var_lo).astype(np.float32)
    inv_vars = (1 / var).astype(np.float32)
    miv = (mean * inv_vars).astype(np.float32)          # DiagGmm::SetMeans after SetInvVars
    gc, nb = orc.compute_gconsts(w, inv_vars, miv)
    assert nb == 0
    return w, mean, var, inv_vars, miv, gc


def _density_loglikes(w, mean, var, x):
    """python/tests/test_diag_gmm.py:345-347 in float64."""
    w, mean, var, x = (a.astype(np.float64) for a in (w, mean, var, x))
    e = np.exp(((x - mean) ** 2 / (-2 * var)).sum(1)) / np.sqrt((var * math.pi * 2).prod(1))
    return np.log(w * e)


@pytest.mark.parametrize("seed", range(5))
def test_gconsts_formula(seed):
    # python/tests/test_diag_gmm.py:45-51 (torch.allclose defaults: rtol 1e-5, atol 1e-8)
    rng = np.random.default_rng(20230414 + seed)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 8)
    w64, mean64, var64 = w.astype(np.float64), mean.astype(np.float64), var.astype(np.float64)
    expected = np.log(w64) - 0.5 * (8 * math.log(2 * math.pi) + np.log(var64).sum(1) + (mean64 ** 2 / var64).sum(1))
    np.testing.assert_allclose(gc, expected, rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("seed", range(5))
def test_log_likelihood_vs_gaussian_density(seed):
    # python/tests/test_diag_gmm.py:327-349: abs(log_likes - expected) < 1e-4
    rng = np.random.default_rng(100 + seed)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 8)
    x = rng.random(8).astype(np.float32)
    m = orc.OModel([0, 10], gc, miv, inv_vars)
    post, ll = orc.component_posteriors(m, 0, x)
    per = _density_loglikes(w, mean, var, x)
    expected = np.log(np.exp(per).sum())
    assert abs(ll - expected) < 1e-4
    # :351-373 per component, :528-553 posteriors == softmax(loglikes) and logsumexp within 1e-4
    got = orc.loglikes(gc, miv, inv_vars, x)
    np.testing.assert_allclose(got, per, rtol=1e-5, atol=1e-4)
    sm = np.exp(per - per.max()); sm /= sm.sum()
    np.testing.assert_allclose(post, sm, rtol=1e-4, atol=1e-6)
    assert abs(orc.logsumexp(got) - ll) < 1e-5


def test_log_likelihoods_matrix_and_preselect():
    # python/tests/test_diag_gmm.py:375-434
    rng = np.random.default_rng(7)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 3)
    X = rng.random((3, 3)).astype(np.float32)
    for i in range(3):
        got = orc.loglikes(gc, miv, inv_vars, X[i])
        np.testing.assert_allclose(got, _density_loglikes(w, mean, var, X[i]), rtol=1e-5, atol=1e-4)
        idx = [0, 1, 3, 8, 7, 8, 3, 2]
        np.testing.assert_allclose(got[idx], _density_loglikes(w, mean, var, X[i])[idx], atol=1e-4)


def test_fma_order_restatement_matches():
    # the MFMA-order chain differs from the reference-order sums only by fp32 rounding
    rng = np.random.default_rng(3)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 32, 40)
    x = (rng.standard_normal(40) * 2).astype(np.float32)
    a = orc.loglikes(gc, miv, inv_vars, x)
    b = orc.loglikes(gc, miv, inv_vars, x, fma_order=True)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=2e-3)


def test_accumulate_from_diag_semantics():
    # python/tests/test_mle_diag_gmm.py:165-252: occ += post, mean += post (x) x, var += post (x) x^2,
    # weight scales the posteriors, returned log-like within 1e-5 of logsumexp; dtype float64 (:48-90)
    rng = np.random.default_rng(11)
    w, mean, var, inv_vars, miv, gc = _rand_gmm(rng, 10, 8)
    m = orc.OModel([0, 10], gc, miv, inv_vars)
    x = rng.random(8).astype(np.float32)
    post, ll = orc.component_posteriors(m, 0, x)
    for weight in (1.0, 0.25):
        acc = orc.OAccs(10, 8, 2)
        got_ll = orc.acc_stats_ali(m, [0, 0, 0], x[None, :], [1], acc, weight=weight)
        assert acc.occ.dtype == np.float64 and acc.mean_acc.dtype == np.float64
        p = (post * np.float32(weight)).astype(np.float32)
        np.testing.assert_array_equal(acc.occ, p.astype(np.float64))
        np.testing.assert_array_equal(acc.mean_acc, np.outer(p, x).astype(np.float32).astype(np.float64))
        np.testing.assert_array_equal(acc.var_acc, np.outer(p, (x * x)).astype(np.float32).astype(np.float64))
        assert abs(got_ll - ll) < 1e-5
        assert acc.trans_acc[1] == 1.0 and acc.trans_acc.sum() == 1.0   # scripts/test_gmm_acc_stats_ali.py:106
        assert acc.total_frames == pytest.approx(weight)


def test_gmm_update_flags():
    # python/tests/test_gmm_update_flags.py:9-39 + csrc/model-common.cc:72-85
    assert orc.augment_gmm_flags(0x2) == 0x7     # v => m => w
    assert orc.augment_gmm_flags(0x1) == 0x5
    assert orc.augment_gmm_flags(0x0) == 0x4     # empty => w
    assert orc.augment_gmm_flags(0xF) == 0xF     # kGmmAll keeps the transition bit


def test_gconst_edge_cases():
    # csrc/diag-gmm.cc:131-146: zero weight -> -inf kept and counted; NaN -> error
    w = np.array([0.0, 1.0], np.float32)
    iv = np.ones((2, 3), np.float32); miv = np.zeros((2, 3), np.float32)
    gc, nb = orc.compute_gconsts(w, iv, miv)
    assert nb == 1 and gc[0] == -np.inf and np.isfinite(gc[1])
    iv[1, 0] = -1.0
    with pytest.raises(orc.OracleError):
        orc.compute_gconsts(w, iv, miv)


def test_threaded_em_pass_driver_counts_every_utterance():
    """orc_em_pass_mt (bench.py's cpu_baseline, variant B): same per-utterance calls on POSIX threads."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from helpers import build
    m, gc, om, ut, cost = build(12, 3, 6, n_utt=9, seed=3, max_phones=4)
    g = dict(ut.graphs)
    g["weight"] = np.where(g["ilabel"] >= 1, g["weight"] + cost[g["ilabel"]], g["weight"]).astype(np.float32)
    for nthr in (1, 3):
        frames, utts, failed, secs = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, num_threads=nthr, acoustic_scale=0.1)
        assert frames == ut.frame_off[-1] and utts == 9 and failed == 0 and secs > 0
    frames, utts, failed, _ = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, first_utt=4, n_utt=3, num_threads=2, acoustic_scale=0.1)
    assert utts == 3 and frames == ut.frame_off[7] - ut.frame_off[4]
    frames, utts, failed, _ = orc.em_pass_mt(om, m.id2pdf, g, ut.frame_off, ut.feats, num_threads=2, budget_seconds=0.0, acoustic_scale=0.1)
    assert utts == 0 and frames == 0


# ---- the literal stored answers the reference holds on this path (csrc/eigen-test.cc, csrc/hash-list-test.cc) ----
LSE_V5 = [0.1, 0.3, 0.2, 0.15, 0.25]                                               # csrc/eigen-test.cc:461-464 -> 1.8119
LSE_V10 = [-0.028933119028806686, -0.8265501260757446, 0.31104734539985657, 0.25977903604507446, 0.18070533871650696,
           0.02222185768187046, -1.4124598503112793, -0.5896500945091248, -0.17299121618270874, -0.6516317129135132]   # :466-473 -> 2.1343
SOFTMAX_V = [0.46589261293411255, 0.5329158902168274, 0.45468050241470337, 0.509181022644043, 0.4529399275779724]   # :642-644
SOFTMAX_EXPECTED = [0.1964813768863678, 0.21010152995586395, 0.19429071247577667, 0.205173522233963, 0.19395282864570618]   # :646-648


def test_logsumexp_stored_vectors():
    # csrc/eigen-test.cc:460-475: EXPECT_NEAR(f, 1.8119, 1e-4); EXPECT_NEAR(f, 2.1343, 1e-4)
    assert abs(orc.logsumexp(LSE_V5) - 1.8119) < 1e-4
    assert abs(orc.logsumexp(LSE_V10) - 2.1343) < 1e-4


def test_softmax_stored_vector():
    # csrc/eigen-test.cc:641-655: EXPECT_NEAR(expected[i], actual[i], 1e-4)
    post, lse = orc.softmax(SOFTMAX_V)
    assert np.abs(post - np.asarray(SOFTMAX_EXPECTED, np.float32)).max() < 1e-4
    assert lse == pytest.approx(math.log(sum(math.exp(v) for v in SOFTMAX_V)), abs=1e-5)
    # the same numbers through the GMM entry points the hot path uses: a 1-dim pdf with zero means / inv_vars has
    # component log-likelihoods == gconsts, so LogLikelihood is LogSumExp(v) and ComponentPosteriors is Softmax(v)
    for v, want in ((LSE_V5, 1.8119), (LSE_V10, 2.1343)):
        g = np.asarray(v, np.float32)
        z = np.zeros((len(v), 1), np.float32)
        m = orc.OModel(np.array([0, len(v)], np.int32), g, z, z)
        ll = orc.loglikes_matrix(m, np.array([[0.7]], np.float32), np.array([0], np.int32))
        assert abs(float(ll[0, 0]) - want) < 1e-4
    g = np.asarray(SOFTMAX_V, np.float32)
    z = np.zeros((5, 1), np.float32)
    post2, _ = orc.component_posteriors(orc.OModel(np.array([0, 5], np.int32), g, z, z), 0, np.array([0.3], np.float32))
    assert np.abs(np.asarray(post2) - np.asarray(SOFTMAX_EXPECTED, np.float32)).max() < 1e-4


@pytest.mark.parametrize("seed,key_mod,val_mod", [(0, 200, 50), (1, 200, 50), (2, 127, 50), (3, 256, 50), (4, 200, 7), (5, 200, 50)])
def test_hash_list_behaviour(seed, key_mod, val_mod):
    """csrc/hash-list-test.cc:19-87 replayed against the oracle's HashList restatement (the decoder's own code): fifty
    find-or-insert operations against a std::map baseline, then 100 rounds of Clear -> SetSize(100 + rand % 100) ->
    re-Insert every element under key + 1 -> Delete, checking after every round that the list holds exactly the
    baseline's pairs and that Find agrees with the baseline on ten random keys.  (key_mod 127 / 256 stand for the
    reference's int16 / char instantiations, whose keys wrap; here keys stay below 2^31.)"""
    rng = np.random.default_rng(1000 + seed)
    h = orc.OHashList()
    h.set_size(200)
    m1 = {}
    for _ in range(50):
        key, val = int(rng.integers(0, key_mod)), int(rng.integers(0, val_mod))
        m1[key] = val
        h.put(key, val)
    assert dict(h.items()) == m1 and len(h.items()) == len(m1)
    for _ in range(100):
        m1 = {k + 1: v for k, v in m1.items()}
        n = h.clear_reinsert(100 + int(rng.integers(0, 100)), shift=1)
        assert n == len(m1)
        items = h.items()
        assert len(items) == len(m1) and dict(items) == m1
        for _ in range(10):
            key = int(rng.integers(0, key_mod + 100))
            assert h.find(key) == m1.get(key)
    h.close()


def test_hash_list_insert_keeps_the_first_value_and_list_is_bucket_ordered():
    """hash-list-inl.h:129-174: Insert on an existing key returns the existing element (value untouched) -- what
    FasterDecoder::ProcessEmitting's `e_found->val != new_tok` test relies on (faster-decoder.cc:213-227); a new
    bucket goes to the END of the list, a key hashing into an occupied bucket goes right after that bucket's last
    element: the iteration order the order-faithful decoder kernels must reproduce."""
    h = orc.OHashList()
    h.set_size(10)
    assert h.insert(3, 30) and h.insert(7, 70) and h.insert(13, 130) and h.insert(5, 50)
    assert not h.insert(7, 71) and h.find(7) == 70
    assert h.items() == [(3, 30), (13, 130), (7, 70), (5, 50)]          # 13 % 10 == 3: behind key 3, ahead of 7
    assert h.find(23) is None and h.find(4) is None
    h.close()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hl_replay(script, backend):
    """Run a harness script (oracle/ref_hashlist_harness.cc's protocol, with the golden file's packed commands) -> answers in the
    golden file's packed form.  backend: an object with set_size / insert / put / find / items / drop / clear_reinsert."""
    import hashlib
    ans = []
    bits = []

    def flush():
        if bits:
            ans.append("I= " + "".join(bits))
            bits.clear()
    for line in script:
        a = line.split()
        if a[0] == "I":
            bits.append("1" if backend.insert(int(a[1]), int(a[2])) else "0")
            continue
        if a[0] == "I*":
            for k in a[2:]:
                bits.append("1" if backend.insert(int(k), int(a[1])) else "0")
            continue
        if a[0] == "S":                      # (commands without an answer do not end a run of Insert answers)
            backend.set_size(int(a[1]))
        elif a[0] == "P":
            backend.put(int(a[1]), int(a[2]))
        elif a[0] == "D":
            backend.drop()
        elif a[0] == "F":
            flush()
            v = backend.find(int(a[1]))
            ans.append("F none" if v is None else f"F {v}")
        elif a[0] == "L":
            flush()
            txt = "L" + "".join(f" {k}:{v}" for k, v in backend.items())
            ans.append("L# %d %s" % (txt.count(":"), hashlib.sha1(txt.encode()).hexdigest()[:16]) if txt.count(":") > 12 else txt)
        elif a[0] == "R":
            flush()
            ans.append(f"R {backend.clear_reinsert(int(a[1]), int(a[2]))}")
        else:
            raise AssertionError(line)
    flush()
    return ans


def test_hash_list_against_answers_recorded_from_the_reference_itself():
    """tests/golden/hashlist_ref.json holds what the REFERENCE's own HashList answered (csrc/hash-list.h compiled as it lies,
    oracle/ref_hashlist_harness.cc; recorded by tests/golden/make_hashlist_golden.py): random Insert / find-or-insert / Find / list
    / clear-and-reinsert mixes, and the decoder's own pattern on state sets larger than the hash size (shared buckets: the list
    order is NOT the insertion order).  The oracle's restatement -- the decoder's own code -- must give the same answers."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "hashlist_ref.json")) as fh:
        gold = json.load(fh)
    assert len(gold["cases"]) >= 16
    shared = 0
    for case in gold["cases"]:
        h = orc.OHashList()
        got = _hl_replay(case["script"], h)
        h.close()
        assert got == case["answers"]
        shared += sum(1 for x in case["answers"] if x.startswith("L# "))
    assert shared > 100


def test_hash_list_against_the_reference_binary_when_it_is_here():
    """Where oracle/_ref/hashlist_ref exists (this container: `make -C oracle ref` compiles the reference's header from
    /root/reference), a fresh random script goes through the reference binary and the oracle side by side."""
    import subprocess
    binary = os.path.join(ROOT, "oracle", "_ref", "hashlist_ref")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/hashlist_ref not built (no /root/reference here)")
    rng = np.random.default_rng(77)
    script = ["S 64"]
    for f in range(60):
        script += ["D", f"S {64 + 8 * f}"] + [f"I {int(k)} {f}" for k in rng.integers(0, 3000, size=int(rng.integers(5, 400)))] + ["L"]
        script += [f"F {int(k)}" for k in rng.integers(0, 3000, size=5)]
    r = subprocess.run([binary], input="\n".join(script) + "\n", capture_output=True, text=True, check=True)
    want = r.stdout.splitlines()

    h = orc.OHashList()
    got = []
    for line in script:
        a = line.split()
        if a[0] == "S":
            h.set_size(int(a[1]))
        elif a[0] == "D":
            h.drop()
        elif a[0] == "I":
            got.append("I 1" if h.insert(int(a[1]), int(a[2])) else "I 0")
        elif a[0] == "F":
            v = h.find(int(a[1]))
            got.append("F none" if v is None else f"F {v}")
        elif a[0] == "L":
            got.append("L" + "".join(f" {k}:{v}" for k, v in h.items()))
    h.close()
    assert got == want
