#!/bin/bash
# Evidence of the round's final kernels: vector-memory probe, occupancy probe, counter passes, bench lines, kernel stats, the one-rank RCCL rehearsal, shares.
set -o pipefail
mkdir -p gpurun_out
timeout -k 5 120 tools/ta_probe 2.4 4 > gpurun_out/ta_probe_4.txt && timeout -k 5 120 tools/ta_probe 2.4 8 > gpurun_out/ta_probe_8.txt || exit 1
bash tools/ta_probe_counters.sh > gpurun_out/ta_cal.txt 2>&1 || { tail gpurun_out/ta_cal.txt; exit 1; }
bash tools/occ_probe.sh > gpurun_out/occ_probe.txt 2>&1 || { tail gpurun_out/occ_probe.txt; exit 1; }
bash tools/r2_evidence_a.sh > gpurun_out/ev_a.log 2>&1 || { tail -20 gpurun_out/ev_a.log; exit 1; }
cp gpurun_out/pmc_per_step.json profiles/pmc_per_step.json   # bench.py reads the counter figures from profiles/
bash tools/r2_evidence_b.sh > gpurun_out/ev_b.log 2>&1 || { tail -20 gpurun_out/ev_b.log; exit 1; }
bash tools/rehearse_rccl_one_rank.sh || { tail gpurun_out/rccl_one_rank.err; exit 1; }
timeout -k 10 300 python tools/share_time.py dragon18520_1080p 512 > gpurun_out/share_headline.txt || exit 1
timeout -k 10 300 python tools/share_time.py dragon18520_4k 256 > gpurun_out/share_4k.txt || exit 1
make -C examples > /dev/null 2>&1
timeout -k 10 100 examples/headless_renderer data/models/TheStanfordDragon18520.obj gpurun_out/headless_1080p.ppm 200 1920 1080 > gpurun_out/headless.txt 2>&1 || exit 1
tail -3 gpurun_out/headless.txt
cat gpurun_out/share_headline.txt gpurun_out/share_4k.txt
tail -12 gpurun_out/ev_b.log | cut -c1-200
