#!/bin/bash
# Round-6 measurement set (one gpurun call; summaries are copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06z; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
DB=$(find $O/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_bench.md | head -12 | cut -c1-220
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python bench.py --steps 8 --warmup 3 --no-variants --no-cpu-baseline --prof-rounds 0 > /dev/null 2> $O/pmc_$c.log
  DB=$(find $O/pmc_$c -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB > $O/pmc_$c.txt; grep -E "ppo_update_persistent|disc_fb|disc_reduce|ia_gemm_tn_side|disc_assemble|rn_merge_seq|disc_fwd" $O/pmc_$c.txt | cut -c1-60,92-
done
for v in P_gp10:6 3_airl_ant_1024x16_mb1024_gp10:6 image_gail_64x16_cnn:3 3_airl_ant_tuned_verbatim:2; do
  n=${v%%:*}; r=${v##*:}
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/variant_profile.py $n $r > $O/kt_$n.log 2>&1
  DB=$(find $O/kt_$n -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$n.md | head -8 | cut -c1-200
done
python tools/ppo_update_timing.py 0 > $O/ppo_timing_P.txt 2>&1; tail -14 $O/ppo_timing_P.txt | cut -c1-300
for v in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim 3_airl_ant_1024x16_mb1024; do python tools/ppo_step_us.py 0 8 $v 2>&1 | tail -1; done > $O/ppo_step_us.txt; cat $O/ppo_step_us.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | cut -c1-260 | head -40
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
