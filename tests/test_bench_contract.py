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
    assert rf["peak"] == 2500.0 and rf["fp32_equivalent"]["peak"] == 157.3 and 0.0 < rf["frac_executed"] <= rf["frac"]
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
    d = _run({}, "--no-cpu-baseline", "--k1", "pdf")
    assert d["dtype"] == "f32" and d["fp32_mfma_line"] is None
    rf = d["roofline"]
    assert rf["peak"] == 157.3 and 0.0 < rf["frac_executed"] <= rf["frac"] < 1.0


@pytest.mark.gpu
def test_bench_collective_path_in_a_one_rank_group():
    d = _run({"KHG_BENCH_FORCE_DIST": "1"}, "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["value"] > 1e6 and d["cpu_baseline"] is None
    assert d["check"]["acc_total_frames"] == d["config"]["frames_per_step"]      # the all-reduce of one rank is the identity
    assert d["allreduce_ms_per_step"] is not None and d["allreduce_ms_per_step"] >= 0.0 and d["allreduce_bytes"] > 0
    # the run says itself how many ranks RCCL saw and what the exchange costs alone (a one-rank RCCL communicator here)
    rc = d["rccl"]
    assert rc["nranks"] == 1 and rc["rank"] == 0 and rc["version_code"] > 20000 and rc["c1_ms_alone"] >= 0.0 and rc["c1_bytes"] == d["allreduce_bytes"]
    assert d["scaling_efficiency_vs_n1_shard"] is None          # only reported for N > 1
