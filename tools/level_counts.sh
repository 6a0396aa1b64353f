#!/bin/bash
# Rays and tests per bounce level (differences of runs with bounce limit 1, 2, 3, ...) and the sq1 counter pass.
set -o pipefail
W=${1:-dragon18520_1080p}; T=${2:-4}
for b in 1 2 3 4 8; do echo "bounce $b"; timeout -k 10 100 python tools/trace_once.py --workload $W --stats --bounce $b --reps 1 --trav $T | tail -1 || exit 1; done
