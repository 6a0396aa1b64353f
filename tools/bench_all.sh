#!/bin/bash
# bench.py lines of the other workloads (no CPU baseline): tools/bench_all.sh <tag>
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/ev_$1
mkdir -p $O
for w in dragon2426_1080p rtiow_1080p dragon18520_4k dragon_field_k8_1080p dragon_field_k23_1080p; do
  steps=10; case $w in dragon18520_4k|dragon_field_k23_1080p) steps=3;; esac
  timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w --steps $steps --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err || exit 1
  python -c "import json,sys; d=json.loads(open('$O/bench_$w.json').read().strip().split('\n')[-1]); print('$w', round(d['value']), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step', 'frac', round(d['roofline']['frac'],3))"
done
