#!/bin/bash
# Round-6 evidence after the last kernel change (K2 score prefetch by 32-byte sectors): GPU suite, rocprofv3 passes, the default bench
# line with the counter summary of the same sources in place, time-boxed sweeps of the K2 paths.
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q > gpurun_out/r6_gpu_tests.log 2>&1; tail -n 2 gpurun_out/r6_gpu_tests.log
bash tools/profile_r6.sh > gpurun_out/r6_profile.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r6 r6 > /dev/null 2>&1
python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
python - > gpurun_out/r6_fuzz_graphs.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("graph fuzz ok:", fuzzlib.fuzz_graphs(ctx, budget=330.0, seed=9702))
PY
python - > gpurun_out/r6_fuzz_parity.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
print("fuzz ok:", fuzzlib.fuzz_parity(ctx, budget=270.0, seed=9701))
PY
tail -n 1 gpurun_out/r6_fuzz_graphs.txt gpurun_out/r6_fuzz_parity.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["roofline"].get("traffic"), d["check"]["ali_mismatch_utts"], d["check"]["oracle_utts"], (d.get("flat_start_line") or {}).get("ms_per_step"))
PY
