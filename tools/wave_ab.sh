#!/bin/bash
# A/B of the wave-synchronous walk of camera batches (RT_WAVE_CAMERA), then the GPU suite.
set -o pipefail
for w in ${WLS:-dragon18520_1080p dragon2426_1080p dragon18520_ground_1080p dragon18520_full_ground_1080p dragon_field_k8_1080p}; do
  for m in 0 1; do
    echo -n "RT_WAVE_CAMERA=$m $w: "
    RT_WAVE_CAMERA=$m timeout -k 10 200 python tools/trace_once.py --workload $w --spp ${SPP:-64} --reps 3 --trav 4 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1 ms/' || exit 1
  done
done
if [ -z "$NOTEST" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/wave_tests.log 2>&1 || { tail -40 gpurun_out/wave_tests.log; exit 1; }
tail -3 gpurun_out/wave_tests.log
fi
