# Round-4 measurements (run through gpurun; summaries are copied to profiles/ by hand):
#  1 kernel trace of the headline command   2 HBM-traffic PMC passes (separate runs)   3 matrix-pipe counters for the
#  discriminator update alone (SQ_VALU_MFMA_BUSY_CYCLES & friends, per MI355X_MICROARCH.md's counter notes)
#  4 kernel traces of variant H (1 024 000-transition rounds: the bandwidth-bound kernels against 8 TB/s) and of the two
#  tuned files   5 phase clocks of the persistent PPO update   6 data-parallel round, stub collectives (row-sharded and
#  replicated)   7 the full bench line   8 the slow parity tests   9 (end of round) the 64-wide epoch kernels' phase
#  clocks + A/B, the closing reduction's A/B, a host profile and the round timelines
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CU|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_ANY|SQ_WAIT_INST_ANY" | cut -c1-160 | sort -u | head -40 > $O/counters_available.txt
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
DB=$(find $O/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_bench.md | head -16
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python bench.py --steps 8 --warmup 3 --no-variants --no-cpu-baseline --prof-rounds 0 > /dev/null 2> $O/pmc_$c.log
  DB=$(find $O/pmc_$c -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB > $O/pmc_$c.txt; grep -E "ppo_update_persistent|disc_fb|disc_reduce|ia_gemm_tn_side|ia_gemm_kernelILi2ELi2ELi1ELi1ELi2|disc_assemble|rn_merge_seq" $O/pmc_$c.txt | cut -c1-60,92-
done
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$n -- python tools/disc_step_bench.py 8 > $O/pmc_$n.log 2>&1
  DB=$(find $O/pmc_$n -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB > $O/pmc_$n.txt 2>> $O/pmc_$n.log; grep -E "disc_fb|disc_reduce|ia_gemm_tn_side|ia_gemm_kernelILi2" $O/pmc_$n.txt | cut -c1-50,92-
done
for v in H_horizon_1024x1000:2 T_gail_half_cheetah_tuned_verbatim:8 3_airl_ant_tuned_verbatim:3 P_mlp64_1024x16:6 P_gp10:6 P_ant_gail_d35_gp10:6; do
  n=${v%%:*}; r=${v##*:}
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/variant_profile.py $n $r > $O/kt_$n.log 2>&1
  DB=$(find $O/kt_$n -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$n.md | head -14
done
python tools/ppo_update_timing.py 0 > $O/ppo_timing_P.txt 2>&1; tail -14 $O/ppo_timing_P.txt | cut -c1-400
python tools/ppo_update_timing.py 0 T_gail_half_cheetah_tuned_verbatim > $O/ppo_timing_T.txt 2>&1; tail -3 $O/ppo_timing_T.txt | cut -c1-400
for v in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim 3_airl_ant_1024x16_mb1024; do python tools/ppo_step_us.py 0 8 $v 2>&1 | tail -1; done > $O/ppo_step_us.txt; cat $O/ppo_step_us.txt
python tools/ppo_epoch_timing.py > $O/ppo_epoch_timing_ll.txt 2>&1; tail -5 $O/ppo_epoch_timing_ll.txt | cut -c1-500
IA_EPOCH_SPLIT=3 python tools/ppo_epoch_timing.py > $O/ppo_epoch_timing_barriers.txt 2>&1; tail -3 $O/ppo_epoch_timing_barriers.txt | cut -c1-500
for m in 3 0; do IA_EPOCH_SPLIT=$m python tools/variant_profile.py P_mlp64_1024x16 24 2>&1 | tail -1 | cut -c1-110; done > $O/mlp64_ab.txt; cat $O/mlp64_ab.txt
for s_ in 0 1; do SIDE=$s_ python tools/disc_step_bench.py 320 2>&1 | grep -E "^fused"; done > $O/disc_side_ab.txt; cat $O/disc_side_ab.txt
python tools/host_profile.py P 10 2>&1 | grep -v "^$" | cut -c1-150 | sed "s#/tmp/code/[^ ]*/repo/##" | head -48 > $O/host_profile_P.txt
python tools/round_timeline.py 8 1 > $O/round_timeline_P.txt 2>&1; python tools/round_timeline.py 8 1 P_gp10 > $O/round_timeline_P_gp10.txt 2>&1
python tools/dp_overhead.py 20 > $O/dp_overhead_sharded.txt 2>&1; head -8 $O/dp_overhead_sharded.txt
IA_DP_ROW_SHARDED=0 python tools/dp_overhead.py 20 > $O/dp_overhead_replicated.txt 2>&1; head -8 $O/dp_overhead_replicated.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | cut -c1-300 | head -24
IA_SLOW_TESTS=1 python -m pytest tests/test_adversarial_gpu.py -q -s -m gpu -k "horizon_rollouts or bc_nature or full_size" > $O/slow_tests.txt 2>&1; grep -E "worst deviation|fraction outside|passed|failed" $O/slow_tests.txt | cut -c1-600
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
