#!/bin/bash
# The vector-memory probe under the L1 counters the trace kernels are measured with: what the counters say for access
# patterns whose lane-loads, lines and misses are known.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ta_cal
mkdir -p $O
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $O/tcp --output-format csv -- $GRAFT_REPO_ROOT/tools/ta_probe 2.4 4 > $O/tcp.log 2>&1 || { tail -5 $O/tcp.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for p in glob.glob("gpurun_out/ta_cal/tcp/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"][:40]})[r["Counter_Name"]] = float(r["Counter_Value"])
loads = 1024 * 256 * 256 * 4
for d in sorted(rows):
    r = rows[d]
    print(d, r["k"], "acc/load %.3f cache_acc/load %.3f tcc_req/load %.3f cyc %.0f" % (r.get("TCP_TOTAL_ACCESSES_sum", 0) / loads, r.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / loads, r.get("TCP_TCC_READ_REQ_sum", 0) / loads, r.get("GRBM_GUI_ACTIVE", 0) / 8))
PY
