#!/bin/bash
# Occupancy experiment: the path kernels with 4 (default), 3 and 2 blocks per CU (unused LDS added to every block).
set -o pipefail
for pad in 0 24576 40960; do
  for w in dragon18520_1080p dragon_field_k8_1080p rtiow_1080p; do
    echo "== RT_LDS_PAD=$pad $w"
    RT_LDS_PAD=$pad timeout -k 10 120 python tools/trace_once.py --workload $w --spp ${SPP:-64} --reps 3 --trav 4 | tail -1 || exit 1
  done
done
