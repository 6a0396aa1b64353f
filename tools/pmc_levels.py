#!/usr/bin/env python3
"""Per-launch view of the counter CSVs written by tools/pmc_passes.sh: one row per k_path / k_fold launch of the
traced call, one column per counter."""
import csv
import glob
import sys

table = {}  # dispatch order -> {counter: value}
names = {}
taken = {}  # (launch, counter) -> the pass file it was taken from
for path in sorted(glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv")):
    with open(path) as fh:
        rows = [r for r in csv.DictReader(fh) if any(k in r["Kernel_Name"] for k in ("k_trace", "k_path", "k_refill", "k_sliced", "k_fold", "k_block", "k_camera"))]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    rank = {d: i for i, d in enumerate(ids)}
    for r in rows:
        i = rank[int(r["Dispatch_Id"])]
        table.setdefault(i, {})
        if (i, r["Counter_Name"]) in taken and taken[(i, r["Counter_Name"])] != path:
            continue  # a counter collected in more than one pass: the first pass's value stands
        taken[(i, r["Counter_Name"])] = path
        table[i][r["Counter_Name"]] = table[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[i] = r["Kernel_Name"][-40:]
cols = sorted({c for t in table.values() for c in t})
print("launch," + ",".join(cols))
for i in sorted(table):
    print(f"{i}," + ",".join(f"{table[i].get(c, 0):.4g}" for c in cols))
