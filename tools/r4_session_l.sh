#!/bin/bash
# K2 at two states per thread (one wave per utterance up to 128 states) against the default one state per thread, same box
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for v in 0 2 4; do
KHG_K2_KS=$v python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('ks=$v', 'step %.2f' % d['ms_per_step'], 'k1 %.2f k2 %.2f' % (k['k1_loglikes'], k['k2_viterbi_dp']), 'll', d['check']['avg_loglike_per_frame'])"
done; done
