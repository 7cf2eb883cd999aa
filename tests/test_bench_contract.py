"""bench.py's output contract: stdout is exactly ONE JSON line with the fields the driver reads (plus `roofline` and
`cpu_baseline`), also when the collective code path runs (KHG_BENCH_FORCE_DIST=1: a one-rank RCCL group on a one-GPU
box -- RCCL's version banner must not reach stdout)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--utts", "2000", "--steps", "2", "--warmup", "1", *flags],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    d = _run({}, "--cpu-baseline-seconds", "3")
    assert KEYS <= set(d), KEYS - set(d)
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["higher_is_better"] is True
    assert d["value"] > 1e6 and d["ms_per_step"] > 0 and d["vs_baseline"] is None and d["dtype"] == "f32 via f16x2s" and d["dtype_note"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0.0 < rf["frac"] < 1.0 and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert rf["peak"] == 2500.0 and rf["fp32_equivalent"]["peak"] == 157.3 and rf["frac"] == rf["frac_executed"] and rf["frac"] <= rf["frac_dense_contract"] < 1.0
    # K2 / K3 against the HBM roofline on the same line
    for k in ("k2", "k3"):
        kk = rf["kernels"][k]
        assert kk["bytes"] > 0 and kk["ms"] > 0 and 0.0 < kk["frac_hbm"] < 1.0 and kk["GBps"] == pytest.approx(kk["bytes"] / kk["ms"] / 1e6)
    # what is derived per parameter version is inside the timed steps
    assert d["kernel_ms_per_step"]["k0s_pack_tiles"] > 0 and d["kernel_ms_per_step"]["k0_model_stats"] > 0
    # the recipe's beams on the same set
    rb = d["recipe_beam_line"]
    assert rb["beam"] == 6.0 and rb["retry_beam"] == 40.0 and rb["value"] > 1e6 and rb["failed_utts"] == 0 and 0 <= rb["retried_utts"] <= rb["fallback_decoder_utts"] <= 2000
    # the reference's per-utterance call pattern at the benchmark's model size
    pc = d["per_call_line"]
    assert pc["utterances"] >= 200 and pc["num_done"] == pc["utterances"] and pc["ali_identical_to_batched"] is True and pc["stats_within_2e-5"] is True
    assert pc["tot_count"] == pc["frames"] and pc["value"] > 1e5
    assert pc["vs_cpu_one_thread"] == pytest.approx(pc["value"] / d["cpu_baseline"]["one_thread_value"])
    # the fp32-MFMA K1 timed beside it, same steps otherwise
    f32 = d["fp32_mfma_line"]
    assert f32["value"] > 1e6 and f32["k1_kernel_ms"] > 0 and 0.0 < f32["roofline_frac"] < 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "frames/s" and cb["sample"]
    assert d["m_step"]["device"]["params_bit_equal_to_host"] is True
    # parity at the bench's own shape, on the record: the utterances the CPU baseline aligned, against K1 + K2 + K3
    ck = d["check"]
    assert ck["oracle_utts"] >= 20 and ck["ali_mismatch_utts"] == 0 and ck["status_mismatch"] == 0 and ck["like_max_rel_err"] < 2e-5
    assert ck["stats_within_2e-5"] is True and ck["trans_acc_equal"] is True and ck["total_frames_equal"] is True


@pytest.mark.gpu
def test_bench_with_the_fp32_mfma_k1():
    d = _run({}, "--no-cpu-baseline", "--k1", "pdf", "--per-call-utts", "0", "--no-recipe-beam-line")
    assert d["dtype"] == "f32" and d["fp32_mfma_line"] is None
    rf = d["roofline"]
    assert rf["peak"] == 157.3 and 0.0 < rf["frac_executed"] == rf["frac"] <= rf["frac_dense_contract"] < 1.0


@pytest.mark.gpu
def test_bench_collective_path_in_a_one_rank_group():
    d = _run({"KHG_BENCH_FORCE_DIST": "1"}, "--no-cpu-baseline", "--per-call-utts", "0", "--no-recipe-beam-line")
    assert d["n_gpus"] == 1 and d["value"] > 1e6 and d["cpu_baseline"] is None
    assert d["check"]["acc_total_frames"] == d["config"]["frames_per_step"]      # the all-reduce of one rank is the identity
    assert d["allreduce_ms_per_step"] is not None and d["allreduce_ms_per_step"] >= 0.0 and d["allreduce_bytes"] > 0
    # the run says itself how many ranks RCCL saw and what the exchange costs alone (a one-rank RCCL communicator here)
    rc = d["rccl"]
    assert rc["nranks"] == 1 and rc["rank"] == 0 and rc["version_code"] > 20000 and rc["c1_ms_alone"] >= 0.0 and rc["c1_bytes"] == d["allreduce_bytes"]
    assert d["scaling_efficiency_vs_n1_shard"] is None          # only reported for N > 1


@pytest.mark.gpu
def test_launch_ladder_gets_a_line_out_after_a_failed_communicator():
    """`python bench.py --gpus 2` from a bare shell starts its ranks itself; when the exchange as asked for cannot form its communicator
    (KHG_BENCH_FAIL_COMM=1: every rank exits 3, in the selftest and in the run) the launcher must fall through to the next rung in
    FRESH children and still print one line, with what failed recorded in it.  Two ranks share GPU 0 here (KHG_BENCH_SHARE_GPU=1)."""
    env = dict(os.environ, KHG_BENCH_SHARE_GPU="1", KHG_BENCH_FAIL_COMM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--utts", "1000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--per-call-utts", "0", "--no-recipe-beam-line", "--no-fp32-line"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 1e5
    fb = d["allreduce_fallback"]
    assert len(fb) == 1 and fb[0]["rc"] != 0 and "selftest" in fb[0]["tried"] and any("KHG_BENCH_FAIL_COMM" in ln for ln in fb[0]["stderr_tail"]), fb


@pytest.mark.gpu
def test_dist_selftest_is_quick_and_checks_the_sum():
    env = dict(os.environ, KHG_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist-selftest"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["selftest"] == "ok" and d["sum"] == d["want"] == 1.0 and d["bytes"] > 1000000 and d["allreduce"] == "khg"


@pytest.mark.gpu
def test_in_process_fallback_when_the_library_communicator_cannot_be_formed():
    """Under the driver's own torch.distributed.run there is no launch ladder: if the library's RCCL communicator cannot be formed
    (KHG_BENCH_FAIL_COMM=2) the ranks agree over the process group and every rank drops to torch.distributed's all-reduce on the
    same block -- the line still comes out, with what happened recorded in it."""
    d = _run({"KHG_BENCH_FORCE_DIST": "1", "KHG_BENCH_FAIL_COMM": "2"}, "--no-cpu-baseline", "--per-call-utts", "0", "--no-recipe-beam-line", "--no-fp32-line")
    assert d["allreduce"] == "torch" and d["rccl"] is None and d["value"] > 1e6
    fb = d["allreduce_fallback_in_process"]
    assert fb["now"] == "--allreduce torch" and "KHG_BENCH_FAIL_COMM=2" in fb["error"]
    assert d["check"]["acc_total_frames"] == d["config"]["frames_per_step"] and d["allreduce_ms_per_step"] is not None
