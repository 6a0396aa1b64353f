#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv) of `bench.py --no-cpu-baseline`: the time the timed steps took on the GPU.

With pipelining the launches of consecutive steps overlap, so per-kernel average durations no longer add up to a step;
what compares with bench.py's HIP-event time is the span from the first launch of the first timed step to the end of
the last fold.  usage: trace_span.py kernel_trace.csv steps launches_per_step  (flat scenes: k_camera + k_refill + k_fold = 3 per chunk of 64
samples per pixel; pointer scenes: k_path per bounce level + k_fold)"""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("k_path", "k_sliced", "k_refill", "k_fold", "k_camera"))]
steps = int(sys.argv[2])
per = int(sys.argv[3]) if len(sys.argv) > 3 else 11
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
timed = rows[-steps * per:]
start = min(int(r["Start_Timestamp"]) for r in timed)
end = max(int(r["End_Timestamp"]) for r in timed)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed)
names = sorted({r["Kernel_Name"][:60] for r in timed})
print(json.dumps({"steps": steps, "launches": len(timed), "span_ms": (end - start) / 1e6, "span_ms_per_step": (end - start) / 1e6 / steps,
                  "sum_of_kernel_durations_ms_per_step": busy / 1e6 / steps, "kernels": names}, indent=1))
