#!/bin/bash
# Long randomised sweeps on the final sources (tests/fuzzlib.py), $1 seconds each (default 1500; seeds move with $2) -> gpurun_out/r6_fuzz_long_*_<seconds>.txt
cd "$GRAFT_REPO_ROOT"
SECS=${1:-1500}; SEED=${2:-0}
for what in parity graphs; do
python - > gpurun_out/r6_fuzz_long_${what}_${SECS}.txt 2>&1 <<PY
import sys
sys.path.insert(0, "tests")
import fuzzlib
from kaldi_hmm_gmm_amd import Context
ctx = Context(0)
f = fuzzlib.fuzz_parity if "$what" == "parity" else fuzzlib.fuzz_graphs
print("$what fuzz ok:", f(ctx, budget=float($SECS), seed=$SEED + (9611 if "$what" == "parity" else 9612)))
PY
tail -n 1 gpurun_out/r6_fuzz_long_${what}_${SECS}.txt
done
python tools/chain_debug.py $((SEED + 707)) $((SECS / 2)) > gpurun_out/r6_chain_debug_long.txt 2>&1; tail -n 2 gpurun_out/r6_chain_debug_long.txt
