#!/bin/bash
# Per-launch durations of one trace call: tools/levels_env.sh <tag> <trav> [workload]   (environment is passed through)
set -o pipefail
TAG=$1; T=$2; W=${3:-dragon18520_1080p}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lv_$TAG -o p -- python3 $R/tools/trace_once.py --workload $W --reps 2 --trav $T > $R/gpurun_out/lv_$TAG.log 2>&1 || exit 1
tail -1 $R/gpurun_out/lv_$TAG.log
python3 $R/tools/levels.py $R/gpurun_out/lv_$TAG/p_kernel_trace.csv 20 | cut -c1-100
