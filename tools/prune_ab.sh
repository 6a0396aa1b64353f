#!/bin/bash
# A/B of the wide walk's distance limit (RT_PRUNE=0 / 1), one trace call per workload: tools/prune_ab.sh
for wl in "dragon18520_1080p 64" "dragon18520_ground_1080p 64" "dragon18520_full_ground_1080p 64" "dragon_field_k8_1080p 64" "dragon_field_k23_1080p 16"; do
  set -- $wl
  for pr in 0 1; do
    echo -n "$1 spp=$2 RT_PRUNE=$pr: "
    RT_PRUNE=$pr RT_DEBUG_LAYOUT=${DBG:-0} timeout -k 10 300 python tools/trace_once.py --workload $1 --spp $2 --reps 4 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel_ms=\([0-9.]*\).*replays=\([0-9]*\)/\1 (\2)/' | tr '\n' ' ' || exit 1
    echo
  done
done
