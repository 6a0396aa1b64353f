#!/usr/bin/env python3
"""Whole-step figures from the counter passes of ONE trace call (tools/pmc_step.sh): HBM bytes, VALU issue utilisation,
active lanes per VALU instruction, summed over every kernel of the call.  Prints one JSON object.
usage: pmc_step.py <pmc dir> <source label>"""
import csv
import glob
import json
import sys

KERNELS = ("k_trace", "k_path", "k_refill", "k_sliced", "k_fold", "k_camera")
tot = {}
for path in sorted(glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv")):
    with open(path) as fh:
        seen = set()
        for r in csv.DictReader(fh):
            if not any(k in r["Kernel_Name"] for k in KERNELS):
                continue
            name = r["Counter_Name"]
            if (path, name) not in seen and name in tot and name in ("SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
                continue  # collected in more than one pass: keep the first pass's
            tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
            seen.add((path, name))
cycles = tot["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
simds = 256 * 4
# The vector-memory path of a CU, measured with tools/ta_probe.hip on this GPU (profiles/r02_ta_probe.txt): lanes that read
# one small record each out of different cache lines are served at 1.98 lines per clock and CU when the lines are in the
# L1, at 0.437 when they come from the L2 (one 128-byte line per miss over the 64-byte-per-clock fill path); the record's
# size (4 / 8 / 16 bytes) does not matter.  TCP_TOTAL_CACHE_ACCESSES counts those line accesses (lanes of a quad that share a
# line count once), TCP_TCC_READ_REQ the misses.  The floor below is the time the counted accesses take at those rates.
L1_HIT_LINES_PER_CLK, L1_MISS_LINES_PER_CLK = 1.98, 0.437
acc, miss = tot.get("TCP_TOTAL_CACHE_ACCESSES_sum"), tot.get("TCP_TCC_READ_REQ_sum")
vmem = None
if acc and miss is not None:
    floor_cycles = ((acc - miss) / L1_HIT_LINES_PER_CLK + miss / L1_MISS_LINES_PER_CLK) / 256.0
    vmem = {"l1_line_accesses_per_step": acc, "l1_miss_requests_per_step": miss, "l1_hit_rate": 1.0 - miss / acc,
            "floor_cycles_per_cu": floor_cycles, "frac_of_vector_memory_roof": floor_cycles / cycles,
            "lines_per_clock_and_cu": acc / 256.0 / cycles,
            "roof_lines_per_clock_and_cu": acc / 256.0 / floor_cycles}
out = {
    # memory-side reads are tallied at 64 B per 128-B request on gfx950: doubled; writes as counted (both in KB)
    "hbm_bytes_per_step": int((2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0),
    "hbm_read_bytes_per_step": int(2.0 * tot["FETCH_SIZE"] * 1024.0),
    "hbm_write_bytes_per_step": int(tot["WRITE_SIZE"] * 1024.0),
    # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (one wave alone issues one per 4 cycles, so it takes
    # two or more waves to reach this rate); kernels of one call run one after the other
    "valu_issue_utilisation": tot["SQ_INSTS_VALU"] * 2.0 / (simds * cycles),
    "active_lanes": tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_INSTS_VALU"],
    "valu_instructions_per_step": tot["SQ_INSTS_VALU"],
    "vmem_read_instructions_per_step": tot.get("SQ_INSTS_VMEM_RD"),
    "ta_busy": tot["TA_TA_BUSY_sum"] / 256.0 / cycles if "TA_TA_BUSY_sum" in tot else None,
    "gpu_cycles_under_counters": cycles,
    "vector_memory_path": vmem,
    "source": sys.argv[2] if len(sys.argv) > 2 else None,
}
print(json.dumps(out))
