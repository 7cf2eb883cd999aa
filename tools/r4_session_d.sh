#!/bin/bash
# round 4, GPU session D: K1 on transcripts that share pdfs (Zipf lexicon) against the independent-phone set: ms and HBM-side traffic
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/zipf_r4; rm -rf $OUT; mkdir -p $OUT
for tr in uniform zipf; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --transcripts $tr > $OUT/bench_$tr.json 2> $OUT/bench_$tr.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --transcripts $tr > /dev/null 2> $OUT/fetch_$tr.log
done
python3 - <<'PY'
import csv, glob, json
for tr in ("uniform", "zipf"):
    d = json.load(open(f"gpurun_out/zipf_r4/bench_{tr}.json"))
    k = d["kernel_ms_per_step"]
    f = glob.glob(f"gpurun_out/zipf_r4/fetch_{tr}/*/*_counter_collection.csv")[0]
    v = {}
    for r in csv.DictReader(open(f)):
        if "k1s_loglikes" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            v[r["Dispatch_Id"]] = v.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    big = [x for x in v.values() if x > 0.1 * max(v.values())]
    print(tr, "step %.2f ms  k1 %.2f  k2 %.2f  k3 %.2f" % (d["ms_per_step"], k["k1_loglikes"], k["k2_viterbi_dp"], k["k3_accumulate"]),
          "| K1 FETCH_SIZE x2: %.1f GB per launch" % (sum(big) / len(big) * 1024 * 2 / 1e9), "|", d["config"]["workload"][-60:], "| check", d["check"].get("avg_loglike_per_frame"))
PY
