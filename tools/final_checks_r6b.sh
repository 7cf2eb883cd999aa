#!/bin/bash
# The side measurements of tools/final_checks_r6.sh again after the last kernel change (K2's score prefetch), without its sweeps
# (tools/final_evidence_r6b.sh ran those) and without the K3 probes (K3 unchanged).
cd "$GRAFT_REPO_ROOT"
bash tools/bench_configs.sh > gpurun_out/r6_bench_configs.txt 2>&1
bash tools/shard_lines.sh > gpurun_out/r6_shard_lines.txt 2>&1
python bench.py --config stress10000x128 --utts 125000 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | tail -1 > gpurun_out/r6_stress_125000_line.json
python tools/hard_model_probe.py 5000 0.01,0.1,0.3 prof > gpurun_out/r6_hard_model_probe.txt 2>&1
python tools/fallback_stress.py > gpurun_out/r6_fallback_stress.txt 2>&1
python tools/chain_debug.py 808 150 > gpurun_out/r6_chain_debug_300s.txt 2>&1
tail -3 gpurun_out/r6_chain_debug_300s.txt; cat gpurun_out/r6_bench_configs.txt gpurun_out/r6_shard_lines.txt; tail -c 600 gpurun_out/r6_stress_125000_line.json
