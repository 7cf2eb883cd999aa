#!/bin/bash
# quick per-kernel counter averages (one bench step per pass): usage tools/pmc_quick.sh "<counters>" [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq_${2:-a}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $1 --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line > /dev/null 2> $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not any(x in k for x in ("k1s_loglikes", "k2_viterbi_dp", "k3_accumulate")):
        continue
    d = acc[k][r["Counter_Name"]]
    d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for k, cs in acc.items():
    print(k[:60], {c: "%.4g" % (sum(v.values()) / len(v)) for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
PY
