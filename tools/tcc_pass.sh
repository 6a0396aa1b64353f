#!/bin/bash
# tools/tcc_pass.sh <workload> <spp>: L2 requests, hits, misses and reads from memory of one trace call, per kernel
export TMPDIR=/tmp
O=gpurun_out/tcc_$1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/a --output-format csv -- python3 tools/trace_once.py --workload $1 --spp $2 --reps 1 > $O.log 2>&1 || { tail -3 $O.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_32B_sum -d $O/b --output-format csv -- python3 tools/trace_once.py --workload $1 --spp $2 --reps 1 > $O.log 2>&1 || { tail -3 $O.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum -d $O/c --output-format csv -- python3 tools/trace_once.py --workload $1 --spp $2 --reps 1 > $O.log 2>&1 || { tail -3 $O.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, re, sys
tot = {}
for path in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_\w+<[^>]*>", r["Kernel_Name"])
        if m:
            tot.setdefault(m.group(0), {})[r["Counter_Name"]] = tot.setdefault(m.group(0), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in sorted(tot.items()):
    print(k, " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
PY
