#!/bin/bash
# tools/time_all.sh: kernel ms (HIP events) of four trace calls of every bench workload, one line each
for w in "dragon18520_1080p 256" "dragon2426_1080p 256" "dragon18520_ground_1080p 128" "dragon18520_full_ground_1080p 64" "dragon_field_k8_1080p 64" "dragon_field_k23_1080p 16" "rtiow_1080p 64"; do
  set -- $w
  echo -n "$1 spp=$2: "
  timeout -k 10 300 python3 tools/trace_once.py --workload $1 --spp $2 --reps 4 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1/' | tr '\n' ' ' || exit 1
  echo
done
