#!/bin/bash
# N = 1 lines at the 2 / 4 / 8-GPU shard sizes of the default workload -> gpurun_out/r6_shard_<n>.json (profiles/r6_shard_lines.json)
cd "$GRAFT_REPO_ROOT"
for n in 50000 25000 12500; do
  python bench.py --utts $n --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line > gpurun_out/r6_shard_$n.json 2> gpurun_out/r6_shard_$n.err
done
python - <<'PY'
import json
out = {}
for n in (50000, 25000, 12500):
    d = json.load(open(f"gpurun_out/r6_shard_{n}.json"))
    k = d["kernel_ms_per_step"]
    out[f"tri5000x64:{n}"] = {"ms_per_step": d["ms_per_step"], "kernel_ms": sum(k.values()), "kernel_ms_per_step": k, "frames": d["config"]["frames_per_step"],
                               "value": d["value"], "command": f"python bench.py --utts {n} --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line",
                               "where": "one MI355X (gpurun box), round 6, final sources"}
    print(n, "%.2f ms" % d["ms_per_step"], {a: round(b, 2) for a, b in k.items()})
json.dump(out, open("gpurun_out/r6_shard_lines.json", "w"), indent=1)
PY
