mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_lookahead.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_t9.log 2>&1 || { tail -40 gpurun_out/r2_t9.log; exit 1; }
tail -3 gpurun_out/r2_t9.log
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread --deselect tests/test_gpu_lookahead.py > gpurun_out/r2_t9b.log 2>&1 || { tail -40 gpurun_out/r2_t9b.log; exit 1; }
tail -3 gpurun_out/r2_t9b.log
make -C examples > /dev/null 2>&1
timeout -k 10 100 examples/headless_renderer data/models/TheStanfordDragon18520.obj gpurun_out/headless_1080p.ppm 200 1920 1080
timeout -k 10 100 examples/headless_renderer data/models/TheStanfordDragon18520.obj gpurun_out/headless_360p.ppm 400 640 360
