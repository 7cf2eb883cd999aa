"""The randomised sweeps and the larger end-to-end check ON THE RECORD (they only existed as manual scripts):
seeded, time-boxed slices of tests/fuzzlib.py under `pytest -m gpu`, and the end-to-end mismatch report written to
profiles/r6_parity_report.json (also gpurun_out/, which is what comes back from the GPU box)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [101, 202])
def test_fuzz_general_graphs_slice(ctx, seed):
    """Status / alignment / words equal to the oracle's FasterDecoder on random general graphs (branches, long skips,
    epsilon-input arcs with words, unreachable finals, empty graphs, too-short utterances, narrow beams, retries,
    max_active) -- every assertion is inside fuzzlib.fuzz_graphs."""
    import fuzzlib
    r = fuzzlib.fuzz_graphs(ctx, budget=8.0, seed=seed)
    assert r["batches"] >= 12 and r["utterances"] >= 60, r
    assert r["fallback"] > 0 and r["retried"] > 0 and r["oracle_failed"] > 0, r      # the slice reaches the order-faithful decoders, retries and failures
    assert r["band_batches"] >= 3, r                                                 # ... and K1's band form + repair on the same random graphs


@pytest.mark.parametrize("seed", [303, 404])
def test_fuzz_parity_slice(ctx, seed):
    """Random shapes (G 1..128, D 1..80, ragged pdfs) and beams: K1 within 1e-5 + 1e-6 B of fp64, K2 bit-exact on
    identical scores, K3 within rtol 2e-4, K4 parameters bit-exact and gconsts <= 4 ulp vs the oracle."""
    import fuzzlib
    r = fuzzlib.fuzz_parity(ctx, budget=9.0, seed=seed)
    assert r["configurations"] >= 4 and r["utterances"] >= 16, r


def test_end_to_end_against_oracle_with_mismatch_report(ctx):
    """K1 -> K2 -> K3 through the C-ABI vs the oracle pipeline, 300 bench-like utterances (600 pdfs x 64 Gaussians x 40 dims),
    beams 200/0 (pruning off) and 6/40 (the recipe's), then scored with a flat-start-like model at beams 6/40 and 2/8 (heavy
    pruning, retries, the order-faithful decoders): a status or alignment mismatch is tolerated only as a
    consequence of fp32 score rounding -- a near-tie between two paths (costs re-scored on the oracle's log-likes within
    1e-3), or a pruning decision at the beam edge, proven by the oracle's decoder reproducing the GPU alignment bit for bit
    from the GPU's scores; statistics within the stated tolerances whenever the alignments agree."""
    import fuzzlib
    rep = fuzzlib.validate_large(ctx, n_utt=300)
    # the same utterances scored with a flat-start-like model: the beam prunes, utterances retry and leave the certified
    # exact-DP path for the order-faithful decoders
    hard = fuzzlib.validate_large(ctx, n_utt=300, beams=((6.0, 40.0), (2.0, 8.0)), flat_noise=0.3)
    assert sum(r["fallback_decoder"] for r in hard["runs"]) >= 20, hard
    rep["runs"] += [dict(r, scoring_model=hard["scoring_model"]) for r in hard["runs"]]
    for d in (os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            with open(os.path.join(d, "r6_parity_report.json"), "w") as fh:
                json.dump(rep, fh, indent=1)
    assert rep["frames"] > 60000
    for run in rep["runs"]:
        # a status (retried / failed) may differ only where the oracle's decoder, fed the GPU's scores, reproduces the GPU's
        # status and alignment: a beam-edge decision on the last bit of a score, never a decoder difference
        assert all(run["status_mismatch_reproduced_by_oracle_decoder_on_gpu_scores"]), run
        assert run["status_mismatches"] <= 0.01 * rep["utterances"], run
        for dlt, expl in zip(run["mismatch_path_cost_deltas"], run["mismatch_reproduced_by_oracle_decoder_on_gpu_scores"]):
            assert expl or (dlt is not None and dlt <= rep["near_tie_bound"]), run
        assert run["alignment_mismatch_rate"] <= 0.01, run
        assert run["max_rel_like_err"] <= 2e-5, run
        if run["alignment_mismatches"] + run["status_mismatches"] == 0:
            assert run["trans_acc_equal"] is True
            assert run["occ_max_err_rel_to_max"] <= 2e-5 and run["mean_acc_max_err_rel_to_max"] <= 2e-5 and run["var_acc_max_err_rel_to_max"] <= 2e-5, run
