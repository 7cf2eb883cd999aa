#!/bin/bash
# A/B of two prebuilt libraries (ab_libs/libkhg_hip_{old,new}.so, built here from two states of the sources): alternating bench lines,
# then a slice of the K2 tests on the new one.
cd "$GRAFT_REPO_ROOT"
line() { timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$1', 'step', round(d['ms_per_step'],2), 'k1', round(k.get('k1_loglikes'),2), 'k2', round(k.get('k2_viterbi_dp'),3), 'k3', round(k.get('k3_accumulate'),2))"; }
for r in 1 2; do
  for v in old new; do cp ab_libs/libkhg_hip_$v.so kaldi_hmm_gmm_amd/libkhg_hip.so; line $v; done
done
cp ab_libs/libkhg_hip_new.so kaldi_hmm_gmm_amd/libkhg_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | tail -n 3
