#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv (device-side gaps of a launch-bound phase).
    python tools/trace_gaps.py <kernel_trace.csv> [name-substring of the phase's first kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "t_gather"
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
if len(starts) < 3:
    raise SystemExit("phase marker not found often enough")
a, b = starts[-2], starts[-1]          # the last complete step
seg = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
gaps.sort()
print(f"{len(seg)} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us; "
      f"gap p50 {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f} us, max {gaps[-1] / 1e3:.1f} us")
