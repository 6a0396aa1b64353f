#!/bin/bash
# tools/ktrace.sh <tag> <launches_per_call> <trace_once.py args...>: kernel trace of trace_once.py, per-launch report of its last call
set -o pipefail
TAG=$1; PER=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 $R/tools/trace_once.py "$@" > $O/run.log 2> $O/run.err || { tail -5 $O/run.err; exit 1; }
cd $R
tail -1 $O/run.log
python tools/ktrace_report.py $(ls $O/*/p_kernel_trace.csv $O/p_kernel_trace.csv 2>/dev/null | head -1) $PER | tee $O/report.txt
