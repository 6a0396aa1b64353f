#!/bin/bash
# A/B of two builds of librt_hip.so on one box: ab/librt_hip_A.so and ab/librt_hip_B.so are copied into place in turn.
# TRAV=<n> tools/ab_so.sh [workloads...]; prints kernel ms of one call (HIP events) and the pipelined bench value.
set -o pipefail
WL=${@:-dragon18520_1080p}
T=${TRAV:-4}
for rep in 1 2; do for v in A B; do
  cp ab/librt_hip_$v.so raytracing_amd/librt_hip.so || exit 1
  for w in $WL; do
    echo -n "$v $w: "
    timeout -k 10 200 python tools/trace_once.py --workload $w --reps 3 --trav $T | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/one call \1 ms/' | tr '\n' ' ' || exit 1
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-counts --workload $w --steps 10 --traversal $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('| bench', round(d['value']), 'Mrays/s', round(d['ms_per_step'],3), 'ms')" || exit 1
  done
done; done
