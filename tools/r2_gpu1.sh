set -e
mkdir -p gpurun_out
echo skip-suite

timeout -k 10 120 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 100 --timeout-method thread > gpurun_out/r2_t1_noalloc.log 2>&1 || { tail -40 gpurun_out/r2_t1_noalloc.log; exit 1; }
tail -3 gpurun_out/r2_t1_noalloc.log
timeout -k 10 200 python bench.py --steps 20 --warmup 2 > gpurun_out/r2_b0.json 2> gpurun_out/r2_b0.err || { tail -20 gpurun_out/r2_b0.err; exit 1; }
cat gpurun_out/r2_b0.json | cut -c1-400
timeout -k 10 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/gloo_probe.py device > gpurun_out/r2_gloo_device.log 2>&1 || echo "gloo device probe failed rc=$?"
tail -12 gpurun_out/r2_gloo_device.log
