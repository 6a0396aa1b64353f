set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_configs.py -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/r2_t2_configs.log 2>&1 || { tail -40 gpurun_out/r2_t2_configs.log; exit 1; }
tail -8 gpurun_out/r2_t2_configs.log
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --deselect tests/test_gpu_configs.py > gpurun_out/r2_t2.log 2>&1 || { tail -40 gpurun_out/r2_t2.log; exit 1; }
tail -3 gpurun_out/r2_t2.log
bash tools/rehearse_n2.sh --steps 5 --warmup 1 --spp 64 > gpurun_out/r2_n2_tile.log 2>&1 || echo "n2 tile rehearsal rc=$?"
tail -1 gpurun_out/r2_n2_tile.log | cut -c1-300
bash tools/rehearse_n2.sh --steps 5 --warmup 1 --spp 64 --split sample > gpurun_out/r2_n2_sample.log 2>&1 || echo "n2 sample rehearsal rc=$?"
tail -1 gpurun_out/r2_n2_sample.log | cut -c1-300
