mkdir -p gpurun_out
for m in busy busy_side; do
  f=gpurun_out/r2_gloo_$m.log
  timeout -k 5 70 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/gloo_probe.py $m > $f 2>&1
  echo "== $m rc=$?"; grep "done in\|finished\|File.*line" $f | head -12
done
exit 0
