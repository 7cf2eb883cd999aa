#!/bin/bash
# same-box A/B of K1's shifted tiles: off / 16-frame shift only / any shift (KHG_K1B_DBG = 32 / 64 / 0)
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
for v in 32 64 0; do
KHG_K1B_DBG=$v python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('dbg=$v', 'step %.2f' % d['ms_per_step'], 'k1 %.2f k2 %.2f' % (k['k1_loglikes'], k['k2_viterbi_dp']), 'exec', round(d['roofline']['executed_cell_fraction'],4))"
done; done
