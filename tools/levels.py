#!/usr/bin/env python3
"""Prints the per-launch durations of the last trace call in a rocprofv3 kernel trace (csv)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -12:]:
    print(r['Kernel_Name'][:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us grid', r['Grid_Size_X'],
          'vgpr', r['VGPR_Count'], 'agpr', r['Accum_VGPR_Count'], 'scratch', r['Scratch_Size'])
