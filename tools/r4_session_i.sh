#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz.py -m gpu -q -x -k "band or loglikes or scale or fuzz or align" 2>&1 | tail -6
for i in 1 2; do
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('step %.2f ms' % d['ms_per_step'], {a: round(b,2) for a,b in k.items()}, 'll', d['check']['avg_loglike_per_frame'], d['roofline']['executed_cell_fraction'])"
done
python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 6 --no-fp32-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['check'])"
