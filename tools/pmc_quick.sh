#!/bin/bash
# tools/pmc_quick.sh <tag> <trace_once args...>: the two SQ counter passes of one trace call, per-launch digest
TAG=$1; shift
O=gpurun_out/pmc_$TAG
export TMPDIR=/tmp
mkdir -p $O
ARGS=("$@")
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$O/$name" --output-format csv -- python3 tools/trace_once.py "${ARGS[@]}" > "$O/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$O/$name.log"; exit 1; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE GRBM_TA_BUSY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
run sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
python tools/pmc_levels.py $O > $O/per_launch.csv && python tools/pmc_report.py < $O/per_launch.csv | tee $O/digest.txt
