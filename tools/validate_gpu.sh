#!/bin/bash
# One GPU call: smoke, the whole GPU suite, then the bench lines and the headline under rocprofv3 --kernel-trace --stats.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_t13.log 2>&1 || { tail -40 gpurun_out/r2_t13.log; exit 1; }
tail -2 gpurun_out/r2_t13.log
bash tools/r2_evidence_b.sh > gpurun_out/ev_b.log 2>&1 || { tail -20 gpurun_out/ev_b.log; exit 1; }
tail -12 gpurun_out/ev_b.log | cut -c1-220
