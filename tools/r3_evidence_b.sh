#!/bin/bash
# Timing evidence of the final kernels, one GPU call (profiles/pmc_per_step.json of the same kernels in place): the bench line,
# the same command under rocprofv3 --kernel-trace --stats with the span of its timed launches, the other workloads, the
# two-rank rehearsals on one card, the one-rank pass through RCCL, what a share of a step costs, the headless frame loop.
set -o pipefail
O=gpurun_out/ev_r03
R=$GRAFT_REPO_ROOT
mkdir -p $O
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cut -c1-200 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-reference-counts > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err || { tail -5 $R/$O/rocprof.err; exit 1; }
cd $R
PER=$(python3 -c "import json; d=json.load(open('$O/bench_under_rocprof.json')); print(3 * ((d['config']['spp_per_step'] + 63) // 64))")
python tools/trace_span.py $O/prof/p_kernel_trace.csv 20 $PER > $O/trace_span.json || exit 1
head -7 $O/trace_span.json
for w in dragon2426_1080p rtiow_1080p dragon18520_ground_1080p dragon18520_full_ground_1080p dragon_field_k8_1080p; do
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
bash tools/rehearse_rccl_one_rank.sh || { tail gpurun_out/rccl_one_rank.err; exit 1; }
timeout -k 10 300 python tools/share_time.py dragon18520_1080p 512 > $O/share_headline.txt || exit 1
timeout -k 10 300 python tools/share_time.py dragon18520_4k 256 > $O/share_4k.txt || exit 1
cat $O/share_headline.txt $O/share_4k.txt
timeout -k 10 100 examples/headless_renderer data/models/TheStanfordDragon18520.obj $O/headless_1080p.ppm 200 1920 1080 > $O/headless.txt 2>&1 || exit 1
tail -3 $O/headless.txt
echo "all timing evidence ok"
