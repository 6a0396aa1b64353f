mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_t7.log 2>&1 || { tail -40 gpurun_out/r2_t7.log; exit 1; }
tail -3 gpurun_out/r2_t7.log
tools/ktrace.sh sep5 12 --spp 64 --reps 3 --camera 1
