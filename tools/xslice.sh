#!/bin/bash
set -o pipefail
W=${1:-dragon18520_1080p}
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-counts --workload $W "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   bench', round(d['value']), round(d['ms_per_step'],3))" || exit 1; }
echo "default"; timeout -k 10 100 python tools/trace_once.py --workload $W --reps 3 --trav 4 | tail -1 | cut -c1-110 || exit 1; b
for cfg in "64 262144 6" "96 262144 6" "128 262144 6" "64 65536 10" "96 65536 10" "128 65536 8"; do set -- $cfg
echo "sliced $1 tail_rays $2 extra rounds $3"; export RT_X_SLICE=$1 RT_X_TAIL_RAYS=$2 RT_X_ROUNDS=$3
timeout -k 10 100 python tools/trace_once.py --workload $W --reps 3 --trav 11 | tail -1 | cut -c1-110 || exit 1; b --traversal 11
done
