set -e
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --workload dragon18520_full_ground_1080p --steps 5 --warmup 1 --cpu-seconds 4 > gpurun_out/r2_b1_pair.json 2> gpurun_out/r2_b1_pair.err || { tail -20 gpurun_out/r2_b1_pair.err; exit 1; }
cut -c1-300 gpurun_out/r2_b1_pair.json
bash tools/rehearse_n2.sh --steps 5 --warmup 1 --spp 64 > gpurun_out/r2_n2_tile.log 2>&1 || echo "n2 tile rehearsal rc=$?"
tail -5 gpurun_out/r2_n2_tile.log | cut -c1-400
bash tools/rehearse_n2.sh --steps 5 --warmup 1 --spp 64 --split sample > gpurun_out/r2_n2_sample.log 2>&1 || echo "n2 sample rehearsal rc=$?"
tail -5 gpurun_out/r2_n2_sample.log | cut -c1-400
