#!/bin/bash
# tools/lane_calibration.sh: what the SQ counters report for kernels whose active lanes per VALU instruction are KNOWN
# (tools/valu_probe with 64 / 32 / 16 / 5 lanes of every wave at work, three instruction mixes: multiply-adds; byte conversion, multiply-add, maximum; reciprocal, multiply-add): the figures tools/pmc_step.py
# derives active lanes from.  Writes gpurun_out/lane_calibration.txt.
export TMPDIR=/tmp
O=gpurun_out/lanecal
mkdir -p $O
: > gpurun_out/lane_calibration.txt
for mix in 0 1 3; do
for lanes in 64 32 16 5; do
    timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/l${lanes}_m${mix} --output-format csv -- tools/valu_probe 2.4 $lanes $mix > $O/l${lanes}_m${mix}.log 2>&1 || { echo "pass failed"; tail -3 $O/l${lanes}_m${mix}.log; exit 1; }
    python3 - $O/l${lanes}_m${mix} $lanes $mix >> gpurun_out/lane_calibration.txt <<'PY'
import csv, glob, sys
tot = {}
for path in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "k_valu" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
i, t, a = tot["SQ_INSTS_VALU"], tot["SQ_THREAD_CYCLES_VALU"], tot["SQ_ACTIVE_INST_VALU"]
print(f"lanes {sys.argv[2]:>2} mix {sys.argv[3]}: THREAD_CYCLES_VALU/INSTS_VALU {t / i:7.3f}  THREAD_CYCLES_VALU/ACTIVE_INST_VALU {t / a:7.3f}  ACTIVE_INST_VALU/INSTS_VALU {a / i:6.3f}  " + " ".join(f"{k}={v:.4g}" for k, v in sorted(tot.items())))
PY
done; done
cat gpurun_out/lane_calibration.txt
