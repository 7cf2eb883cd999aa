#!/bin/bash
# round 4, GPU session B: the whole GPU suite, the default bench line, the rocprofv3 passes -- all on one build
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4_gpu_tests.log
python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
bash tools/profile_r4.sh > gpurun_out/r4_profile.log 2>&1
tail -5 gpurun_out/r4_gpu_tests.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench_default.json"))
print("%.2f ms/step %.1f M" % (d["ms_per_step"], d["value"] / 1e6), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()}, d["roofline"])
print(d["check"]); print(d["cpu_baseline"])
PY
