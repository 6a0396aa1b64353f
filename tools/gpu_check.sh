#!/bin/bash
# One GPU call: smoke, the GPU test suite, timings of the three main workloads, and a per-launch kernel trace.
# Every step runs under its own time limit and the chain stops at the first failure.
set -o pipefail
TAG=${1:-x}
TRAV=${TRAV:-4}
R=$GRAFT_REPO_ROOT
timeout -k 5 40 python tools/trace_once.py --workload dragon2426_1080p --width 320 --height 180 --spp 2 --reps 1 --trav $TRAV &&
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 60 --timeout-method thread > gpurun_out/t_$TAG.log 2>&1 &&
for w in dragon18520_1080p dragon2426_1080p rtiow_1080p; do timeout -k 10 60 python tools/trace_once.py --workload $w --reps 3 --trav $TRAV | tail -1 || exit 1; done &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/tools/trace_once.py --reps 3 --trav $TRAV > $R/gpurun_out/prof_$TAG.log 2>&1
