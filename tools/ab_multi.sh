#!/bin/bash
# A/B/... of several builds of librt_hip.so on one box: tools/ab_multi.sh "<trace_once args>" ab/librt_hip_X.so ...
# Each build is copied into place in turn (two passes); prints the kernel ms of the last of 3 calls.
ARGS=$1; shift
cp raytracing_amd/librt_hip.so /tmp/librt_hip_keep.so
for rep in 1 2; do for so in "$@"; do
  cp $so raytracing_amd/librt_hip.so || exit 1
  echo -n "$so: "; timeout -k 10 120 python tools/trace_once.py $ARGS --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1 ms/'
done; done
cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so
