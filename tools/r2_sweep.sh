mkdir -p gpurun_out
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python tools/trace_once.py --spp 64 --reps 3 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1 ms/'; }
run RT_SLICE=128
run RT_TAIL_RAYS=1048576
run RT_TAIL_RAYS=4194304
run RT_TAIL_RAYS=16777216
run RT_SLICE=64 RT_EXTRA_ROUNDS=4
run RT_SLICE=64 RT_EXTRA_ROUNDS=4 RT_TAIL_RAYS=1048576
run RT_SLICE=256
run RT_SLICE=256 RT_TAIL_RAYS=1048576
run RT_SLICE=96 RT_EXTRA_ROUNDS=3 RT_TAIL_RAYS=2097152
run RT_CHUNK_SAMPLES=32
