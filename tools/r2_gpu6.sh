mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_t4.log 2>&1 || { tail -40 gpurun_out/r2_t4.log; exit 1; }
tail -3 gpurun_out/r2_t4.log
for t in 11 12; do echo -n "trav $t: "; timeout -k 10 120 python tools/trace_once.py --spp 64 --reps 3 --trav $t | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1 ms/'; done
for p in 8 32 64; do echo -n "trav 12 period $p: "; RT_PERIOD=$p timeout -k 10 120 python tools/trace_once.py --spp 64 --reps 3 --trav 12 | tail -1 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1 ms/'; done
