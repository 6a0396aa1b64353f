#!/bin/bash
# State check after a restore: the whole GPU suite, then the default bench line.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_t11.log 2>&1 || { tail -40 gpurun_out/r2_t11.log; exit 1; }
tail -3 gpurun_out/r2_t11.log
timeout -k 10 300 python bench.py > gpurun_out/r2_b11.json 2> gpurun_out/r2_b11.err || { tail -20 gpurun_out/r2_b11.err; exit 1; }
cat gpurun_out/r2_b11.json
