#!/bin/bash
# A/B of several builds of librt_hip.so on one box: every ab/librt_hip_<name>.so is copied into place in turn.
# tools/ab_variants.sh "<workload> <spp>" ... ; prints the kernel ms (HIP events) of four calls each.  REPS=<n> rounds.
set -o pipefail
cp raytracing_amd/librt_hip.so ab/.keep_librt_hip.so
WLS=("$@")
for rep in $(seq 1 ${REPS:-2}); do for so in ab/librt_hip_*.so; do
  v=${so#ab/librt_hip_}; v=${v%.so}
  cp $so raytracing_amd/librt_hip.so || exit 1
  for w in "${WLS[@]}"; do
    name=${w% *}; spp=${w#* }
    echo -n "$v $name spp=$spp: "
    timeout -k 10 200 python tools/trace_once.py --workload $name --spp $spp --reps 4 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1/' | tr '\n' ' ' || exit 1
    echo
  done
done; done
cp ab/.keep_librt_hip.so raytracing_amd/librt_hip.so
