#!/bin/bash
# round 4, GPU session A: bench contract tests, the N = 1 lines of the 8-GPU shard sizes, config #5 at its per-GPU shard size
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_bench_contract.py -m gpu -q -x 2>&1 | tail -5
for n in 50000 25000 12500; do
  python bench.py --utts $n --no-cpu-baseline --no-fp32-line > gpurun_out/r4_shard_$n.json 2> gpurun_out/r4_shard_$n.err
done
python bench.py > gpurun_out/r4_bench_5.json 2> gpurun_out/r4_bench_5.err
python bench.py --config stress10000x128 --utts 125000 --cpu-baseline-seconds 10 > gpurun_out/r4_stress_125000.json 2> gpurun_out/r4_stress_125000.err
python - <<'PY'
import json
for f in ["r4_shard_50000", "r4_shard_25000", "r4_shard_12500", "r4_bench_5", "r4_stress_125000"]:
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "%.2f ms/step" % d["ms_per_step"], {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()}, d["roofline"]["frac"], d["roofline"].get("frac_executed"))
        print("   check:", {k: v for k, v in d["check"].items() if k != "note"})
    except Exception as ex:
        print(f, "FAILED", ex)
PY
tail -3 gpurun_out/r4_stress_125000.err
