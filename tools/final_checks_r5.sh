#!/bin/bash
# Round-end evidence on the final sources: the default bench line (with cpu_baseline, check, per_call_line, recipe_beam_line), the other
# configs, the shard-size lines, config #5 at its 8-GPU shard size, and time-boxed runs of the randomised sweeps (tests/fuzzlib.py) --
# outputs under gpurun_out/, summaries copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
bash tools/bench_configs.sh > gpurun_out/r5_bench_configs.txt 2>&1
bash tools/r5_shard_lines.sh > gpurun_out/r5_shard_lines.txt 2>&1
python bench.py --config stress10000x128 --utts 125000 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | tail -1 > gpurun_out/r5_stress_125000_line.json
python - > gpurun_out/r5_fuzz_parity.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("fuzz ok:", fuzzlib.fuzz_parity(ctx, budget=600.0, seed=9501))
PY
python - > gpurun_out/r5_fuzz_graphs.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("graph fuzz ok:", fuzzlib.fuzz_graphs(ctx, budget=600.0, seed=9502))
PY
python - > gpurun_out/r5_validate_large.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print(json.dumps(fuzzlib.validate_large(ctx, n_utt=400), indent=1))
PY
tail -2 gpurun_out/r5_fuzz_parity.txt gpurun_out/r5_fuzz_graphs.txt; cat gpurun_out/r5_bench_configs.txt gpurun_out/r5_shard_lines.txt; tail -c 600 gpurun_out/r5_stress_125000_line.json
