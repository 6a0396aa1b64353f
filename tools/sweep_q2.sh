#!/bin/bash
# With the slice quorum in place: hand-over threshold (kernel ms, last of 3 calls).
set -o pipefail
for w in "dragon18520_1080p --spp 64" "dragon2426_1080p --spp 64" "dragon_field_k8_1080p --spp 16" "dragon_field_k23_1080p --spp 8" "dragon18520_bare_4k --spp 64" "dragon_field_k8_1080p --spp 64"; do
  for t in 4194304 8388608 12582912 16777216 25165824 33554432; do
    echo -n "$w tail=$t: "
    RT_TAIL_RAYS=$t timeout -k 10 150 python tools/trace_once.py --workload $w --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\) Mrays.s=\([0-9.]*\).*/\1 ms \2 Mrays\/s/' || exit 1
  done
done
