#!/bin/bash
# A/B of bench.py settings: tools/bench_ab.sh "<args A>" "<args B>" [workloads...]; prints value and ms per step.
A=$1; B=$2; shift 2
WL=${@:-dragon18520_1080p}
for rep in 1 2; do
  for args in "$A" "$B"; do
    for w in $WL; do
      timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-counts --workload $w $args 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$args', d['config']['workload'], round(d['value']), round(d['ms_per_step'],3))" || exit 1
    done
  done
done
