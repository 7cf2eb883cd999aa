#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 --kernel-trace CSV: the timeline of the LAST bench step (from the last
k1 launch on), every kernel with its start offset, duration and the gap since the previous kernel's end.
usage: gap_report.py <dir with *_kernel_trace.csv> [k1 kernel name prefix]"""
import csv, glob, sys
d = sys.argv[1]
k1 = sys.argv[2] if len(sys.argv) > 2 else "k1s_loglikes"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(k1))
# the step before the last K1 as well: [prev K1, last K1) is one full step
k1s = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(k1)]
a, b = (k1s[-2], k1s[-1]) if len(k1s) > 1 else (k1s[-1], len(rows))
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; busy = 0; gaps = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  gap {gap / 1e6:7.3f}  {r['Kernel_Name'][:70]}")
    busy += e - s; gaps += max(gap, 0); prev_end = max(prev_end, e)
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e6 if b < len(rows) else (prev_end - t0) / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, gaps {gaps / 1e6:.3f} ms")
