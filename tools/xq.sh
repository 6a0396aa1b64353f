#!/bin/bash
set -o pipefail
for W in dragon18520_1080p dragon2426_1080p rtiow_1080p; do
for t in 4 9; do
echo -n "$W trav $t: "; timeout -k 10 100 python tools/trace_once.py --workload $W --reps 3 --trav $t | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/one call \1 ms/' | tr '\n' ' ' || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-counts --workload $W --traversal $t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('| bench', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done; done
