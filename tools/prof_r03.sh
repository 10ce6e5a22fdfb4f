set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03
# 1. kernel trace of the headline command (short)
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > gpurun_out/r03/kt_bench.json 2> gpurun_out/r03/kt.log
DB=$(find gpurun_out/r03/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB gpurun_out/r03/kernel_stats_bench.md | head -25
# 2. PMC passes (separate; kernel-trace only beside --pmc)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/r03/pmc_$c -- python bench.py --steps 8 --warmup 3 --no-variants --no-cpu-baseline --prof-rounds 0 > /dev/null 2> gpurun_out/r03/pmc_$c.log
  DB=$(find gpurun_out/r03/pmc_$c -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB > gpurun_out/r03/pmc_$c.txt; grep -E "ppo_update_persistent|disc_fb|disc_fwd|disc_bwd|disc_reduce|ia_gemm_kernelILi2ELi2ELi1ELi1ELi2|disc_assemble|rn_merge_seq" gpurun_out/r03/pmc_$c.txt | cut -c1-60,92-
done
# 3. variants: default 32x32 discriminator, tuned GAIL verbatim
for v in P_disc32 T_gail_half_cheetah_tuned_verbatim P_gp10; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03/kt_$v -- python tools/variant_profile.py $v 8 > gpurun_out/r03/kt_$v.log 2>&1
  DB=$(find gpurun_out/r03/kt_$v -name "*results.db" | head -1); python tools/rocpd_stats.py $DB gpurun_out/r03/kernel_stats_$v.md | head -12
done
find gpurun_out/r03 -name "*.db" -size +20M -delete; du -sh gpurun_out/r03
