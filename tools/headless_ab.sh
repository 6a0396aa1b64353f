#!/bin/bash
make -C examples > /dev/null 2>&1
for rep in 1 2; do for v in A B; do
  cp ab/librt_hip_$v.so raytracing_amd/librt_hip.so || exit 1
  echo -n "$v: "; timeout -k 10 100 examples/headless_renderer data/models/TheStanfordDragon18520.obj gpurun_out/headless_1080p.ppm 200 1920 1080 2>&1 | grep frames
  echo -n "$v 360p: "; timeout -k 10 100 examples/headless_renderer data/models/TheStanfordDragon18520.obj gpurun_out/headless_360p.ppm 400 640 360 2>&1 | grep frames
done; done
