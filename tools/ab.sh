#!/bin/bash
# A/B of the traversal variants on the main workloads (one process per line; kernel ms from HIP events)
for trav in ${TRAVS:-0 2 1 3}; do
  echo "== traversal $trav"
  python tools/trace_once.py --spp ${SPP:-2} --reps 2 --trav $trav | tail -1
  python tools/trace_once.py --workload dragon2426_1080p --reps 2 --trav $trav | tail -1
  python tools/trace_once.py --workload rtiow_1080p --spp 8 --reps 2 --trav $trav | tail -1
done
