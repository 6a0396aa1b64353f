#!/bin/bash
# Sliced walk on the big flat scenes: slice length x tail threshold x extra rounds (kernel ms of the last of 3 calls).
set -o pipefail
for w in "dragon18520_full_ground_1080p --spp 16" "dragon_field_k8_1080p --spp 16" "dragon_field_k23_1080p --spp 8" "dragon18520_1080p --spp 64"; do
  for cfg in "128 65536 8" "96 65536 12" "64 65536 16" "64 32768 24" "96 32768 16" "128 32768 12" "128 131072 6" "160 65536 8"; do
    set -- $cfg
    echo -n "$w slice=$1 tail=$2 extra=$3: "
    RT_SLICE=$1 RT_TAIL_RAYS=$2 RT_EXTRA_ROUNDS=$3 timeout -k 10 120 python tools/trace_once.py --workload $w --trav 11 --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1 ms/' || exit 1
  done
done
