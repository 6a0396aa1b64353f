mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_t8.log 2>&1 || { tail -40 gpurun_out/r2_t8.log; exit 1; }
tail -3 gpurun_out/r2_t8.log
for a in "--camera 1" "--camera 2"; do echo -n "dragon 512spp $a: "; timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-reference-counts 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"; done
for w in rtiow_1080p dragon18520_full_ground_1080p dragon_field_k8_1080p dragon2426_1080p; do echo -n "$w: "; timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-reference-counts 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config']['spp_per_step'])"; done
