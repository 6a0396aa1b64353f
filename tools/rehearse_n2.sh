#!/bin/bash
# Rehearses bench.py's N = 2 path on ONE GPU: two ranks share the card, gloo instead of RCCL (which refuses that).
# tools/rehearse_n2.sh [bench.py args...]
set -o pipefail
export RT_BENCH_BACKEND=gloo RT_BENCH_WATCHDOG=150
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --no-cpu-baseline --no-reference-counts "$@"
