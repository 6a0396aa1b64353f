#!/bin/bash
# Round evidence in one GPU call: bench line, the same command under rocprofv3 --kernel-trace --stats, the span of the
# timed launches, and the counter passes of one trace call.  tools/evidence.sh <tag>
set -o pipefail
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev_$TAG
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
cd $R
python tools/trace_span.py $O/prof/p_kernel_trace.csv 10 > $O/trace_span.json || exit 1
tools/pmc_passes.sh $O/pmc --reps 1 --trav 4 > $O/pmc.log 2>&1 || exit 1
python tools/pmc_levels.py $O/pmc > $O/pmc_per_launch.csv || exit 1
python tools/pmc_report.py < $O/pmc_per_launch.csv > $O/pmc_digest.txt || exit 1
tail -1 $O/bench.json | cut -c1-200; cat $O/trace_span.json | head -8; cat $O/pmc_digest.txt
