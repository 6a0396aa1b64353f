#!/bin/bash
# Every bench workload under the automatic choice and under the forced sliced / re-fill walks (kernel ms, last of 3 calls).
set -o pipefail
for w in "dragon18520_1080p --spp 64" "dragon2426_1080p --spp 64" "dragon18520_full_ground_1080p --spp 16" "dragon18520_4k --spp 8" "dragon_field_k8_1080p --spp 16" "dragon_field_k23_1080p --spp 8" "rtiow_1080p --spp 16" "plumbing_400x225 --spp 64"; do
  for trav in 0 11 7; do
    echo -n "$w trav=$trav: "
    timeout -k 10 150 python tools/trace_once.py --workload $w --trav $trav --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\) Mrays.s=\([0-9.]*\).*/\1 ms \2 Mrays\/s/' || exit 1
  done
done
