mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_camera_stage.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_t3_cam.log 2>&1 || { tail -40 gpurun_out/r2_t3_cam.log; exit 1; }
tail -3 gpurun_out/r2_t3_cam.log
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --deselect tests/test_gpu_camera_stage.py > gpurun_out/r2_t3.log 2>&1 || { tail -40 gpurun_out/r2_t3.log; exit 1; }
tail -3 gpurun_out/r2_t3.log
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r2_b2.json 2> gpurun_out/r2_b2.err || { tail -20 gpurun_out/r2_b2.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r2_b2.json'));print(d['value'],d['ms_per_step'],d['rays_per_step'],d['incl_d2h'])"
timeout -k 10 300 python bench.py --steps 640 --warmup 4 --spp 16 --no-cpu-baseline --no-extras --no-reference-counts > gpurun_out/r2_b2_spp16.json 2> gpurun_out/r2_b2_spp16.err || { tail -20 gpurun_out/r2_b2_spp16.err; exit 1; }
cut -c1-200 gpurun_out/r2_b2_spp16.json
