#!/bin/bash
# A/B of the wide tree against the compact tree (RT_WIDE=0) on the bench workloads, one trace call each: tools/wide_ab.sh
for wl in "dragon18520_1080p 64" "dragon18520_1080p 512" "dragon2426_1080p 64" "dragon18520_ground_1080p 64" "dragon18520_full_ground_1080p 64" "dragon_field_k8_1080p 64"; do
  set -- $wl
  for wide in 0 1; do
    echo "== $1 spp=$2 RT_WIDE=$wide"
    RT_WIDE=$wide RT_DEBUG_LAYOUT=1 timeout -k 10 300 python tools/trace_once.py --workload $1 --spp $2 --reps 3 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
  done
done
