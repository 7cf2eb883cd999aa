#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "acc_stats or golden" 2>&1 | tail -3
for tr in uniform zipf; do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-line --transcripts $tr 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$tr', 'step %.2f ms' % d['ms_per_step'], {a: round(b,2) for a,b in k.items()}, 'll', d['check']['avg_loglike_per_frame'], d['check'].get('trans_acc_equal'))"
done
