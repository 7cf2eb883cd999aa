#!/bin/bash
# round 4, final evidence on the final sources: GPU suite, default bench line, rocprofv3 passes, other configs, time-boxed fuzz runs
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4_gpu_tests.log
python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
bash tools/profile_r4.sh > gpurun_out/r4_profile.log 2>&1
bash tools/final_checks_r4.sh > gpurun_out/r4_final_checks.log 2>&1
tail -3 gpurun_out/r4_gpu_tests.log
tail -12 gpurun_out/r4_final_checks.log
