"""Turn the rocprofv3 output of tools/profile_r1.sh (gpurun_out/prof_r1/) into the committed summaries:

  profiles/r1_kernel_stats.csv      kernel-trace --stats rows (our kernels first)
  profiles/r1_pmc_summary.json      per-kernel per-launch PMC averages (+ the gfx950 corrections)
  profiles/r1_bench_under_rocprof.json   the bench line printed under the profiler
  profiles/README.md                tables of the above

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]": FETCH_SIZE and WRITE_SIZE come
from separate --pmc passes, are reported in KB, and on gfx950 FETCH_SIZE counts wide (16 B/lane)
streaming reads at exactly half their bytes -> doubled here; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r1")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r1"
OURS = ("k1_loglikes", "k1p_", "k1b_", "k1h_", "k1s_", "k0b_", "k0h_", "k0s_", "k2_viterbi", "k3_", "k4_", "k0_pack", "c1_", "rocprim", "rccl", "nccl")


def short(name):
    if "rocprim" in name:      # K3's stable (pdf, frame) radix sort (hipcub::DeviceRadixSort) -- and torch's own scans
        return "rocprim::" + ("radix_sort_onesweep" if "radix_sort" in name else name.split("detail::")[-1][:40])
    return name.split("(")[0].replace("void ", "") if any(o in name for o in OURS) else name[:60]


def one(pattern):
    f = glob.glob(os.path.join(SRC, pattern))
    if not f:
        raise SystemExit(f"missing {pattern} under {SRC}")
    return max(f, key=os.path.getmtime)      # gpurun MERGES runs into gpurun_out/: rocprofv3 names files by pid, take the newest


def counters(sub):
    """-> {kernel: {counter: [values per dispatch]}}"""
    out = {}
    with open(one(f"{sub}/*/*_counter_collection.csv")) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if not any(o in k for o in OURS):
                continue
            out.setdefault(short(k), {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            out[short(k)][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: list(v.values()) for c, v in d.items()} for k, d in out.items()}


def mean(v):
    """Average over the MAIN launches of a kernel: a kernel that is also launched as a small side job (a warm-up on a few frames, a
    per-model pass) would otherwise pull every per-launch figure down.  Dispatches under a tenth of the largest are left out."""
    v = [x for x in v if x >= 0.1 * max(v)] if v else v
    return sum(v) / len(v) if v else 0.0


def trace_durations():
    """-> {kernel: [ns per dispatch]} from the kernel trace itself (the stats file only has the average over all dispatches)"""
    out = {}
    with open(one("trace/*/*_kernel_trace.csv")) as fh:
        for r in csv.DictReader(fh):
            out.setdefault(short(r["Kernel_Name"]), []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


stats = []
with open(one("trace/*/*_kernel_stats.csv")) as fh:
    for r in csv.DictReader(fh):
        stats.append(r)
stats.sort(key=lambda r: (0 if any(o in r["Name"] for o in OURS) else 1, -float(r["TotalDurationNs"])))
with open(os.path.join(DST, f"{TAG}_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in stats[:16]:
        w.writerow(r)

bench_line = None
with open(os.path.join(SRC, "bench_trace.json")) as fh:
    for line in fh:
        if line.startswith("{"):
            bench_line = json.loads(line)
if bench_line:
    with open(os.path.join(DST, f"{TAG}_bench_under_rocprof.json"), "w") as fh:
        json.dump(bench_line, fh, indent=1)

fetch = counters("pmc_fetch")
write = counters("pmc_write")
mfma = counters("pmc_mfma")
import hashlib
_h = hashlib.sha256()
_d = os.path.join(ROOT, "kaldi_hmm_gmm_amd", "csrc")
for _fn in sorted(os.listdir(_d)):            # same identity bench.py computes (csrc_sha): which kernels the profile belongs to
    if _fn.endswith((".hip", ".inc")):            # the device code and its launch code; the host classes do not touch the kernels
        with open(os.path.join(_d, _fn), "rb") as _fh:
            _h.update(_fn.encode() + b"\0" + _fh.read())
durs = trace_durations()
_sha = _h.hexdigest()[:16]
_sha_file = os.path.join(SRC, "csrc_sha.txt")      # written on the GPU box by the profile script: the sources that were PROFILED
if os.path.exists(_sha_file):
    _sha = open(_sha_file).read().strip() or _sha
summary = {"command": f"tools/profile_{TAG}.sh", "csrc_sha": _sha, "kernels": {}}
if bench_line:
    nb = bench_line["roofline"]["launches_per_step"]
    summary["utterances_per_launch"] = bench_line["config"]["utterances"] / nb
    summary["frames_per_launch"] = bench_line["config"]["frames_per_step"] / nb
for k in sorted(set(fetch) | set(write) | set(mfma)):
    e = {}
    if k in fetch:
        kb = mean(fetch[k].get("FETCH_SIZE", []))
        e["FETCH_SIZE_KB"] = kb
        e["read_bytes_corrected"] = kb * 1024.0 * 2.0          # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    if k in write:
        kb = mean(write[k].get("WRITE_SIZE", []))
        e["WRITE_SIZE_KB"] = kb
        e["write_bytes"] = kb * 1024.0
    if "read_bytes_corrected" in e and "write_bytes" in e:
        e["traffic_bytes"] = e["read_bytes_corrected"] + e["write_bytes"]
    if k in mfma:
        busy = mean(mfma[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        gui = mean(mfma[k].get("GRBM_GUI_ACTIVE", []))
        e["SQ_VALU_MFMA_BUSY_CYCLES"] = busy
        e["GRBM_GUI_ACTIVE"] = gui
        for cn in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16"):
            if cn in mfma[k]:
                e[cn] = mean(mfma[k][cn])
        # busy cycles are summed over the 1024 SIMDs, GUI_ACTIVE over the 8 XCDs
        e["mfma_busy_frac"] = (busy / 1024.0) / (gui / 8.0) if gui else None
    if k in durs:
        main = [x for x in durs[k] if x >= 0.1 * max(durs[k])]
        e["calls"] = len(durs[k])
        e["main_calls"] = len(main)
        e["avg_ms"] = mean(durs[k]) / 1e6          # over the main launches (see mean())
    summary["kernels"][k] = e
with open(os.path.join(DST, f"{TAG}_pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)

lines = [f"# Round {TAG[1:]} rocprofv3 summaries (MI355X, gfx950)", "",
         f"Produced by `tools/profile_{TAG}.sh` on the GPU box and `tools/summarize_prof.py` here: one",
         "`rocprofv3 --kernel-trace --stats` run of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`",
         "(the default workload: 100 000 utterances, one launch of each kernel per step), then three separate",
         "`--pmc` runs of the same command (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE).", "",
         f"## Kernel trace ({TAG}_kernel_stats.csv)", "", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
for r in stats[:10]:
    lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {r['Percentage']} |")
lines += ["", "Kernels that also run as a much smaller side launch (warm-up calls, per-model passes): average over the main launches, from the kernel trace:", ""]
for k, v in durs.items():
    main = [x for x in v if x >= 0.1 * max(v)]
    if len(main) != len(v) and any(o in k for o in OURS):
        lines.append(f"* `{k}`: {len(main)} main launches, {mean(v) / 1e6:.3f} ms each ({len(v) - len(main)} side launches, {sum(x for x in v if x < 0.1 * max(v)) / max(1, len(v) - len(main)) / 1e6:.3f} ms each)")
lines += ["", f"## PMC, per launch ({TAG}_pmc_summary.json)", "",
          "| kernel | FETCH_SIZE KB | reads, corrected x2 (GB) | WRITE_SIZE (GB) | traffic (GB) | MFMA busy frac |", "|---|---|---|---|---|---|"]
for k, e in summary["kernels"].items():
    if "FETCH_SIZE_KB" not in e:
        continue
    mb = e.get("mfma_busy_frac")
    lines.append(f"| `{k}` | {e['FETCH_SIZE_KB']:.0f} | {e['read_bytes_corrected'] / 1e9:.2f} | {e.get('write_bytes', 0) / 1e9:.2f} | "
                 f"{e.get('traffic_bytes', 0) / 1e9:.2f} | {'' if mb is None else f'{mb:.3f}'} |")
if bench_line:
    rf = bench_line["roofline"]
    lines += ["", f"Bench line under the profiler ({TAG}_bench_under_rocprof.json): K1 ({rf['kernel']}) {rf['kernel_ms']:.2f} ms per launch by HIP events, "
              f"{rf['achieved']:.1f} TFLOP/s = {rf['frac']:.3f} of the {rf['peak']:g} TFLOP/s MFMA peak of its arithmetic."]
notes = os.path.join(DST, f"{TAG}_notes.md")
if os.path.exists(notes):
    lines += ["", open(notes).read().rstrip()]
with open(os.path.join(DST, "README.md" if TAG == "r1" else f"{TAG}_README.md"), "w") as fh:
    fh.write("\n".join(lines) + "\n")
print("\n".join(lines))
