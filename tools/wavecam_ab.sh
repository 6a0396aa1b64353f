#!/bin/bash
# tools/wavecam_ab.sh: one trace call with and without the wave-synchronous camera round (RT_WAVE_CAMERA) in the wide flow
for wl in "dragon18520_1080p 256" "dragon2426_1080p 256" "dragon18520_ground_1080p 128"; do
  set -- $wl
  for wc in 1 0; do
    echo -n "$1 spp=$2 RT_WAVE_CAMERA=$wc: "
    RT_WAVE_CAMERA=$wc timeout -k 10 200 python3 tools/trace_once.py --workload $1 --spp $2 --reps 4 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1/' | tr '\n' ' ' || exit 1
    echo
  done
done
