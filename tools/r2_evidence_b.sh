# bench lines, the headline under rocprofv3 --kernel-trace --stats, the other workloads, the 2-rank rehearsals
mkdir -p gpurun_out/ev_r02
O=gpurun_out/ev_r02
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cut -c1-250 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-reference-counts > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err || { tail -5 $R/$O/rocprof.err; exit 1; }
cd $R
ls $O/prof
for w in dragon2426_1080p rtiow_1080p dragon18520_full_ground_1080p dragon_field_k8_1080p; do
  timeout -k 10 500 python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err || { tail $O/bench_$w.err; exit 1; }
  cut -c1-160 $O/bench_$w.json
done
timeout -k 10 500 python bench.py --workload dragon_field_k23_1080p --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_dragon_field_k23_1080p.json 2> $O/bench_k23.err || { tail $O/bench_k23.err; exit 1; }
cut -c1-160 $O/bench_dragon_field_k23_1080p.json
timeout -k 10 500 python bench.py --workload dragon18520_4k --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_dragon18520_4k.json 2> $O/bench_4k.err || { tail $O/bench_4k.err; exit 1; }
cut -c1-160 $O/bench_dragon18520_4k.json
bash tools/rehearse_n2.sh --steps 10 --warmup 2 > $O/n2_tile.log 2>&1 || echo "n2 tile rc=$?"
tail -1 $O/n2_tile.log | cut -c1-200
bash tools/rehearse_n2.sh --steps 10 --warmup 2 --split sample > $O/n2_sample.log 2>&1 || echo "n2 sample rc=$?"
tail -1 $O/n2_sample.log | cut -c1-200
