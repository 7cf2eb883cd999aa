#!/bin/bash
# Round-end evidence on the final sources: the default bench line (with cpu_baseline), the other configs, and time-boxed runs of the
# randomised sweeps (tests/fuzzlib.py) -- outputs under gpurun_out/, summaries copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT"
bash tools/bench_configs.sh > gpurun_out/r4_bench_configs.txt 2>&1
python bench.py --config mono100x8 --utts 100000 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_bench_mono100x8_100k.json
python - > gpurun_out/r4_fuzz_parity_f16x2s.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("fuzz ok:", fuzzlib.fuzz_parity(ctx, budget=240.0, seed=9001))
PY
python - > gpurun_out/r4_fuzz_graphs_f16x2s.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("graph fuzz ok:", fuzzlib.fuzz_graphs(ctx, budget=240.0, seed=9002))
PY
python - > gpurun_out/r4_validate_large.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print(json.dumps(fuzzlib.validate_large(ctx, n_utt=400), indent=1))
PY
tail -2 gpurun_out/r4_fuzz_parity_f16x2s.txt gpurun_out/r4_fuzz_graphs_f16x2s.txt; cat gpurun_out/r4_bench_configs.txt; tail -c 1500 gpurun_out/r4_bench_default.json
