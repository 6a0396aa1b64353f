#!/usr/bin/env python3
"""Summarises the counter CSVs written by tools/pmc_passes.sh for the trace kernel (k_trace)."""
import csv
import glob
import json
import sys

out = {}
for path in sorted(glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv")):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if not any(k in row["Kernel_Name"] for k in ("k_trace", "k_path", "k_fold")):
                continue
            out.setdefault(row["Counter_Name"], 0.0)
            out[row["Counter_Name"]] += float(row["Counter_Value"])
            out["_vgpr"] = row.get("VGPR_Count")
            out["_grid"] = row.get("Grid_Size")
print(json.dumps(out, indent=1, sort_keys=True))
