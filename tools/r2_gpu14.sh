#!/bin/bash
# Per-sample buffers by owned-tile slot: the whole GPU suite, one headline call, and the shares of a step.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_t14.log 2>&1 || { tail -40 gpurun_out/r2_t14.log; exit 1; }
tail -2 gpurun_out/r2_t14.log
timeout -k 10 120 python tools/trace_once.py --spp 64 --reps 3 --trav 4 | tail -1 || exit 1
timeout -k 10 300 python tools/share_time.py dragon18520_1080p 512 > gpurun_out/share_r2_14.txt || exit 1
cat gpurun_out/share_r2_14.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-reference-counts --no-extras > gpurun_out/r2_b14.json 2> gpurun_out/r2_b14.err || { tail gpurun_out/r2_b14.err; exit 1; }
cut -c1-200 gpurun_out/r2_b14.json
