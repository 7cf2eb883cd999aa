#!/bin/bash
# kernel trace of the Zipf-transcript bench (which bucket kernel slows down under skew)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/zipf_trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --transcripts zipf > $OUT/bench.json 2> $OUT/log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/zipf_trace/t/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:70], r["Calls"], "%.3f ms" % (float(r["AverageNs"]) / 1e6))
PY
