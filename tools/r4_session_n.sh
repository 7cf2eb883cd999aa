#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for v in 32 0 32 0; do
KHG_K1B_DBG=$v python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --beam 10 --retry-beam 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('beam 10/40 dbg=$v: step %.2f ms' % d['ms_per_step'], {a: round(b,2) for a,b in k.items()}, d['check']['avg_loglike_per_frame'])"
done
