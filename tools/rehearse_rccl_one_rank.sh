#!/bin/bash
# bench.py under torch.distributed.run with ONE rank taken through the collectives of a multi-rank run (RT_FORCE_DIST=1:
# RCCL communicator, barrier, reduce of the image, all-reduce of the timings) -- what a box with one GPU can rehearse.
RT_FORCE_DIST=1 RT_BENCH_WATCHDOG=240 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-reference-counts --no-extras \
  > gpurun_out/rccl_one_rank.json 2> gpurun_out/rccl_one_rank.err
