#!/bin/bash
# k_sliced: walking lanes at or below which a wave ends its slice early (kernel ms, last of 3 calls).
set -o pipefail
for w in "dragon18520_1080p --spp 64" "dragon2426_1080p --spp 64" "dragon_field_k8_1080p --spp 16" "dragon_field_k23_1080p --spp 8"; do
  for cfg in "0 128" "8 128" "16 128" "24 128" "32 128" "16 192" "24 256" "32 256"; do
    set -- $cfg
    echo -n "$w quorum=$1 slice=$2: "
    RT_SLICE_QUORUM=$1 RT_SLICE=$2 timeout -k 10 150 python tools/trace_once.py --workload $w --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\) Mrays.s=\([0-9.]*\).*/\1 ms \2 Mrays\/s/' || exit 1
  done
done
