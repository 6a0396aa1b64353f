#!/bin/bash
# Hardware-counter passes for the trace kernel, one rocprofv3 run per pass (at most 8 SQ, 4 TCC, 4 TCP,
# 2 TA, 2 GRBM counters in one pass).  Usage: tools/pmc_passes.sh <outdir> <trace_once.py args...>
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
run() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$out/$name" --output-format csv -- python3 tools/trace_once.py "${ARGS[@]}" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; exit 1; }
}
ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE GRBM_TA_BUSY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
echo "all passes ok"
