#!/bin/bash
# The vector-memory probe (plain and under the L1 counters) and the whole-step counter passes with the L1 pass.
set -o pipefail
mkdir -p gpurun_out
timeout -k 5 120 tools/ta_probe 2.4 4 > gpurun_out/ta_probe_4.txt && timeout -k 5 120 tools/ta_probe 2.4 8 > gpurun_out/ta_probe_8.txt || exit 1
bash tools/ta_probe_counters.sh > gpurun_out/ta_cal.txt 2>&1 || { tail gpurun_out/ta_cal.txt; exit 1; }
bash tools/r2_evidence_a.sh > gpurun_out/ev_a.log 2>&1 || { tail -20 gpurun_out/ev_a.log; exit 1; }
tail -5 gpurun_out/ev_a.log | cut -c1-300
