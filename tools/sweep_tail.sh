#!/bin/bash
# Last round of the sliced walk finished by k_sliced itself (0) or by the re-fill kernel (1), over extra rounds.
set -o pipefail
for w in "dragon18520_full_ground_1080p --spp 16" "dragon_field_k8_1080p --spp 16" "dragon_field_k23_1080p --spp 8" "dragon18520_1080p --spp 64" "dragon2426_1080p --spp 64"; do
  for cfg in "1 131072 0" "1 262144 0" "1 524288 0" "1 1048576 0" "1 2097152 0" "1 4194304 0" "1 1048576 -2" "1 524288 2"; do
    set -- $cfg
    echo -n "$w refill_tail=$1 tail=$2 extra=$3: "
    RT_REFILL_TAIL=$1 RT_TAIL_RAYS=$2 RT_EXTRA_ROUNDS=$3 timeout -k 10 120 python tools/trace_once.py --workload $w --trav 11 --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1 ms/' || exit 1
  done
done
