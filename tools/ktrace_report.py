#!/usr/bin/env python3
"""Per-launch durations of the LAST traced call in a rocprofv3 kernel trace of tools/trace_once.py (csv), in launch order:
usage: ktrace_report.py kernel_trace.csv launches_per_call"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("k_path", "k_sliced", "k_refill", "k_fold", "k_camera", "k_trace", "k_block"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2])
last = rows[-per:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += b - a
    print(f"{(a - t0) / 1e3:10.1f} us  +{(b - a) / 1e3:9.1f} us  {r['Kernel_Name'][:50]}")
print(f"sum {tot / 1e3:.1f} us, span {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
