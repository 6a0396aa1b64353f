#!/bin/bash
# tools/ab_fetch.sh <workload> <spp>: HBM bytes read and written by one trace call (FETCH_SIZE / WRITE_SIZE passes) and its
# kernel time, for every build ab/librt_hip_<name>.so copied into place in turn.
export TMPDIR=/tmp
WL=$1; SPP=$2
cp raytracing_amd/librt_hip.so ab/.keep_librt_hip.so
for so in ab/librt_hip_*.so; do
  v=${so#ab/librt_hip_}; v=${v%.so}
  cp $so raytracing_amd/librt_hip.so || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    O=gpurun_out/abfetch_${v}_$c
    timeout -s KILL 200 rocprofv3 --pmc $c -d $O --output-format csv -- python3 tools/trace_once.py --workload $WL --spp $SPP --reps 1 > $O.log 2>&1 || { echo "pass failed"; tail -3 $O.log; exit 1; }
    python3 - $O $v $c <<'PY'
import csv, glob, sys
tot = {}
for path in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-44:]
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
mul = 2.0 if sys.argv[3] == "FETCH_SIZE" else 1.0   # reads are tallied at 64 B per 128-B request on this GPU
print(sys.argv[2], sys.argv[3], " ".join(f"{k}={v * mul * 1024 / 1e9:.2f}GB" for k, v in sorted(tot.items()) if v * mul * 1024 > 5e7))
PY
  done
  echo -n "$v kernel_ms: "; timeout -k 10 200 python tools/trace_once.py --workload $WL --spp $SPP --reps 3 | sed 's/.*kernel_ms=\([0-9.]*\).*/\1/' | tr '\n' ' '; echo
done
cp ab/.keep_librt_hip.so raytracing_amd/librt_hip.so
