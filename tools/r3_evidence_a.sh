#!/bin/bash
# Counter evidence of the final kernels, one GPU call: whole-step figures of every bench workload (-> gpurun_out/pmc_per_step.json,
# stamped with a hash of the kernel sources; copy it to profiles/ before the bench runs), the per-launch digest of one
# 64-spp call of the headline, L2 figures, the issue-rate probe and the lane calibration.
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/pmc_per_step.json
for w in "dragon18520_1080p 64" "dragon2426_1080p 64" "rtiow_1080p 64" "dragon18520_ground_1080p 64" "dragon18520_full_ground_1080p 64" "dragon18520_4k 16" "dragon_field_k8_1080p 64" "dragon_field_k23_1080p 16"; do
  set -- $w
  tools/pmc_step.sh $1 $2 > gpurun_out/pmc_step_$1.log 2>&1 || { tail -5 gpurun_out/pmc_step_$1.log; exit 1; }
  echo "counters $1 ok"
done
tools/pmc_quick.sh r03_headline64 --spp 64 --reps 1 > gpurun_out/r03_pmc_digest.txt 2>&1 || { tail gpurun_out/r03_pmc_digest.txt; exit 1; }
cp gpurun_out/pmc_r03_headline64/per_launch.csv gpurun_out/r03_pmc_per_launch.csv
cut -c1-230 gpurun_out/r03_pmc_digest.txt
for w in dragon18520_1080p dragon18520_ground_1080p dragon18520_full_ground_1080p; do
  tools/tcc_pass.sh $w 64 > gpurun_out/r03_tcc_$w.txt 2>&1 || { tail gpurun_out/r03_tcc_$w.txt; exit 1; }
done
timeout -k 5 120 tools/valu_probe 2.4 > gpurun_out/r03_valu_probe.txt || exit 1
timeout -k 10 300 tools/lane_calibration.sh > /dev/null || exit 1
cp gpurun_out/lane_calibration.txt gpurun_out/r03_lane_calibration.txt
echo "all counter evidence ok"
