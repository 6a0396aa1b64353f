#!/bin/bash
# Automatic choice against the forced re-fill walk at the bench's chunk sizes (kernel ms, last of 3 calls).
set -o pipefail
for w in "dragon18520_full_ground_1080p --spp 64" "dragon18520_full_ground_1080p --spp 32" "dragon_field_k8_1080p --spp 64" "dragon_field_k23_1080p --spp 64" "dragon18520_4k --spp 23" "dragon18520_1080p --spp 64"; do
  for trav in 0 7; do
    echo -n "$w trav=$trav: "
    timeout -k 10 200 python tools/trace_once.py --workload $w --trav $trav --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\) Mrays.s=\([0-9.]*\).*/\1 ms \2 Mrays\/s/' || exit 1
  done
done
