# Round-5 measurement set in one gpurun call (summaries are copied to profiles/ by hand):
#  1 the GPU suite   2 kernel trace of the headline command   3 HBM-traffic PMC passes (separate runs)   4 kernel traces of
#  P_mlp64_1024x16, image GAIL, P_gp10   5 phase clocks (32-wide persistent update, 64-wide epoch kernel)   6 per-step times
#  7 host profile + round timeline of config P and of the generic-VecEnv pair   8 data-parallel stub rounds (both forms)
#  9 the full bench line   10 the full-size parity tests' measured deviations
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt | cut -c1-200
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
DB=$(find $O/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_bench.md | head -16
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python bench.py --steps 8 --warmup 3 --no-variants --no-cpu-baseline --prof-rounds 0 > /dev/null 2> $O/pmc_$c.log
  DB=$(find $O/pmc_$c -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB > $O/pmc_$c.txt; grep -E "ppo_update_persistent|disc_fb|disc_reduce|ia_gemm_tn_side|disc_assemble|rn_merge_seq" $O/pmc_$c.txt | cut -c1-60,92-
done
for v in P_mlp64_1024x16:6 image_gail_64x16_cnn:3 P_gp10:6; do
  n=${v%%:*}; r=${v##*:}
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/variant_profile.py $n $r > $O/kt_$n.log 2>&1
  DB=$(find $O/kt_$n -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$n.md | head -12
done
python tools/ppo_update_timing.py 0 > $O/ppo_timing_P.txt 2>&1; tail -14 $O/ppo_timing_P.txt | cut -c1-400
python tools/ppo_epoch_timing.py > $O/ppo_epoch_timing.txt 2>&1; tail -3 $O/ppo_epoch_timing.txt | cut -c1-500
for v in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim 3_airl_ant_1024x16_mb1024; do python tools/ppo_step_us.py 0 8 $v 2>&1 | tail -1; done > $O/ppo_step_us.txt; cat $O/ppo_step_us.txt
python tools/host_profile.py P 10 2>&1 | grep -v "^$" | cut -c1-150 | sed "s#/tmp/code/[^ ]*/repo/##" | head -48 > $O/host_profile_P.txt
python tools/host_profile.py image_gail_64x16_cnn 3 2>&1 | grep -v "^$" | cut -c1-150 | head -48 > $O/host_profile_image.txt
python tools/round_timeline.py 8 1 > $O/round_timeline_P.txt 2>&1
for v in P_stagger_arrays_1024 P_generic_vecenv_1024 P_stagger_arrays_1024 P_generic_vecenv_1024; do python tools/variant_profile.py $v 24 2>&1 | tail -1 | cut -c1-110; done > $O/generic_pair.txt; cat $O/generic_pair.txt
python tools/host_profile.py P_generic_vecenv_1024 10 2>&1 | grep -v "^$" | cut -c1-150 | head -40 > $O/host_profile_generic.txt
python tools/dp_overhead.py 20 > $O/dp_overhead_sharded.txt 2>&1; head -8 $O/dp_overhead_sharded.txt
IA_DP_ROW_SHARDED=0 python tools/dp_overhead.py 20 > $O/dp_overhead_replicated.txt 2>&1; head -8 $O/dp_overhead_replicated.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | cut -c1-300 | head -30
python -m pytest tests/test_adversarial_gpu.py -q -s -m gpu -k "horizon_rollouts or full_size" > $O/full_size.txt 2>&1; grep -E "worst deviation|passed|failed" $O/full_size.txt | cut -c1-400
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
