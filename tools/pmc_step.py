#!/usr/bin/env python3
"""Whole-step figures from the counter passes of ONE trace call (tools/pmc_step.sh): HBM bytes, VALU issue utilisation,
active lanes per VALU instruction, summed over every kernel of the call.  Prints one JSON object.
usage: pmc_step.py <pmc dir> <source label> <samples per pixel of the call>

Active lanes: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU.  Both counters weigh an instruction by the issue slots it takes
(a reciprocal or a 64-bit operation takes several), so their ratio is the lanes at work per slot; dividing the first by
the plain instruction count instead overstates it (65-107 "lanes" on kernels with divisions in them).  Checked on kernels
with a KNOWN number of lanes at work, three instruction mixes (tools/lane_calibration.sh): 64.000 / 32.000 / 16.000 /
5.001 reported for 64 / 32 / 16 / 5."""
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
            if (path, name) not in seen and name in tot and name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
                continue  # collected in more than one pass: keep the first pass's
            tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
            seen.add((path, name))
cycles = tot["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
simds = 256 * 4
MEASURED_ISSUE_PEAK = 0.39
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
    # issue slots taken per SIMD and clock (a plain fp32 instruction takes one slot), and that rate against the most this
    # GPU was measured to issue: 0.39 slots per SIMD and clock with 8 waves of independent multiply-adds per SIMD, 0.33 with
    # 4 (tools/valu_probe.hip); kernels of one call run one after the other
    "valu_slots_per_simd_and_clock": tot["SQ_ACTIVE_INST_VALU"] / (simds * cycles),
    "valu_issue_utilisation": tot["SQ_ACTIVE_INST_VALU"] / (simds * cycles) / MEASURED_ISSUE_PEAK,
    "active_lanes": tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"],
    "valu_instructions_per_step": tot["SQ_INSTS_VALU"],
    "valu_slots_per_step": tot["SQ_ACTIVE_INST_VALU"],
    "spp": int(sys.argv[3]),
    "vmem_read_instructions_per_step": tot.get("SQ_INSTS_VMEM_RD"),
    "ta_busy": tot["TA_TA_BUSY_sum"] / 256.0 / cycles if "TA_TA_BUSY_sum" in tot else None,
    "gpu_cycles_under_counters": cycles,
    "vector_memory_path": vmem,
    "source": sys.argv[2] if len(sys.argv) > 2 else None,
}
print(json.dumps(out))
