#!/bin/bash
# round 4, GPU session E: K3 work items (pdfs far above the average get more slices): parity tests, then uniform / Zipf bench lines
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_dist.py -m gpu -q -x -k "acc_stats or em or golden or shard or dist" 2>&1 | tail -8
for tr in uniform zipf; do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-line --transcripts $tr 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$tr', 'step %.2f ms' % d['ms_per_step'], {a: round(b,2) for a,b in k.items()}, 'll', d['check']['avg_loglike_per_frame'])"
done
python3 bench.py --config stress10000x128 --utts 20000 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --transcripts zipf 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('stress zipf 20000', 'step %.2f ms' % d['ms_per_step'], {a: round(b,2) for a,b in k.items()})"
