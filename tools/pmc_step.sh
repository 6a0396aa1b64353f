#!/bin/bash
# tools/pmc_step.sh <workload> <spp> [camera]: counter passes (one rocprofv3 run each) of one trace call of <spp> samples per pixel;
# merges the whole-step figures into gpurun_out/pmc_per_step.json under workloads[<workload>], next to a hash of the kernel
# sources they were measured on (bench.py drops figures measured on other sources than the ones it loads)
WL=$1; SPP=$2; CAM=${3:-0}
O=gpurun_out/pmcstep_${WL}_${SPP}
export TMPDIR=/tmp
mkdir -p $O
run() { name=$1; shift; timeout -s KILL 300 rocprofv3 --pmc "$@" -d "$O/$name" --output-format csv -- python3 tools/trace_once.py --workload $WL --spp $SPP --reps 1 --camera $CAM > "$O/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$O/$name.log"; exit 1; }; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE TA_TA_BUSY_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python tools/pmc_step.py $O "rocprofv3 --pmc passes of tools/trace_once.py --workload $WL --spp $SPP (tools/pmc_step.sh)" $SPP > $O/step.json || exit 1
python - <<PY
import json, os, sys
sys.path.insert(0, ".")
import bench
p = "gpurun_out/pmc_per_step.json"
stamp = bench.source_stamp()
d = json.load(open(p)) if os.path.exists(p) else {}
if d.get("measured_on") != stamp:   # figures of other kernels do not mix with these
    d = {"measured_on": stamp, "workloads": {}}
d["workloads"]["$WL"] = json.load(open("$O/step.json"))
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
print("$WL", d["workloads"]["$WL"])
PY
