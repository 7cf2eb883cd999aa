#!/bin/bash
# Round-6 evidence on the final sources: the default bench line (with cpu_baseline, check, per_call_line, recipe_beam_line, flat_start_line),
# the other configs, the shard-size lines, config #5 at its 8-GPU shard size, the order-faithful decoders' probes and time-boxed runs of the
# randomised sweeps (tests/fuzzlib.py) -- outputs under gpurun_out/, summaries copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
bash tools/bench_configs.sh > gpurun_out/r6_bench_configs.txt 2>&1
bash tools/shard_lines.sh > gpurun_out/r6_shard_lines.txt 2>&1
python bench.py --config stress10000x128 --utts 125000 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | tail -1 > gpurun_out/r6_stress_125000_line.json
python tools/hard_model_probe.py 5000 0.01,0.1,0.3 prof > gpurun_out/r6_hard_model_probe.txt 2>&1
python tools/fallback_stress.py > gpurun_out/r6_fallback_stress.txt 2>&1
python tools/k3_mixed_probe.py 20000 > gpurun_out/r6_k3_mixed_probe.txt 2>&1
python tools/chain_debug.py 606 300 > gpurun_out/r6_chain_debug_300s.txt 2>&1
python - > gpurun_out/r6_fuzz_parity.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("fuzz ok:", fuzzlib.fuzz_parity(ctx, budget=600.0, seed=9601))
PY
python - > gpurun_out/r6_fuzz_graphs.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("graph fuzz ok:", fuzzlib.fuzz_graphs(ctx, budget=600.0, seed=9602))
PY
python - > gpurun_out/r6_validate_large.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print(json.dumps(fuzzlib.validate_large(ctx, n_utt=400), indent=1))
PY
tail -2 gpurun_out/r6_fuzz_parity.txt gpurun_out/r6_fuzz_graphs.txt; tail -3 gpurun_out/r6_chain_debug_300s.txt; cat gpurun_out/r6_bench_configs.txt gpurun_out/r6_shard_lines.txt; tail -c 600 gpurun_out/r6_stress_125000_line.json
