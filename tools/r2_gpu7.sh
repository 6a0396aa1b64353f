mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_t5.log 2>&1 || { tail -40 gpurun_out/r2_t5.log; exit 1; }
tail -3 gpurun_out/r2_t5.log
tools/ab_multi.sh "--spp 64" ab/librt_hip_A.so ab/librt_hip_D.so
tools/ab_multi.sh "--workload rtiow_1080p --spp 64" ab/librt_hip_A.so ab/librt_hip_D.so
tools/ab_multi.sh "--workload dragon2426_1080p --spp 16 --trav 3" ab/librt_hip_A.so ab/librt_hip_D.so
tools/ab_multi.sh "--workload dragon2426_1080p --spp 16 --trav 2" ab/librt_hip_A.so ab/librt_hip_D.so
