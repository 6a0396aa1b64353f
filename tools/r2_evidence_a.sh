# counter passes: whole-step figures for the bench workloads, per-round digest of one 64-spp chunk of the headline
mkdir -p gpurun_out
rm -f gpurun_out/pmc_per_step.json
tools/pmc_step.sh dragon18520_1080p 512 || exit 1
tools/pmc_step.sh dragon18520_1080p 16 || exit 1
tools/pmc_step.sh dragon2426_1080p 512 || exit 1
tools/pmc_step.sh rtiow_1080p 128 || exit 1
tools/pmc_step.sh dragon18520_full_ground_1080p 64 || exit 1
tools/pmc_step.sh dragon_field_k8_1080p 64 || exit 1
tools/pmc_step.sh dragon_field_k23_1080p 64 || exit 1
tools/pmc_quick.sh r02_headline64 --spp 64 --reps 1 > gpurun_out/r02_pmc_digest.txt 2>&1
cat gpurun_out/r02_pmc_digest.txt | cut -c1-220
