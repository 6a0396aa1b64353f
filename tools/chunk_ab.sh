#!/bin/bash
# tools/chunk_ab.sh: time of one trace call against the samples per chunk (RT_CHUNK_SAMPLES): the workspace is
# proportional to it.  Three repetitions each, the last two count.
for wl in "dragon18520_1080p 256" "dragon18520_full_ground_1080p 128" "dragon_field_k8_1080p 128"; do
  set -- $wl
  for cs in 16 32 64; do
    echo "== $1 spp=$2 RT_CHUNK_SAMPLES=$cs"
    RT_CHUNK_SAMPLES=$cs timeout -k 10 200 python3 tools/trace_once.py --workload $1 --spp $2 --reps 3 | cut -c1-110 || exit 1
  done
done
