#!/bin/bash
# Slice length of the sliced walk (with the re-fill hand-over) per scene: kernel ms, last of 3 calls.
set -o pipefail
for w in "dragon18520_full_ground_1080p --spp 16" "dragon18520_4k --spp 8" "dragon_field_k8_1080p --spp 16" "dragon18520_1080p --spp 64" "dragon2426_1080p --spp 64"; do
  for sl in 32 48 64 96 128 192; do
    echo -n "$w slice=$sl: "
    RT_SLICE=$sl timeout -k 10 150 python tools/trace_once.py --workload $w --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\) Mrays.s=\([0-9.]*\).*/\1 ms \2 Mrays\/s/' || exit 1
  done
done
