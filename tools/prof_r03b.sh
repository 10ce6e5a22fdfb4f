# Round-3 end-of-round measurements (after the wide-row discriminator tiles, the block-form PPO bodies and the one-tower
# 64-wide epoch kernel): kernel traces of the headline command and of the two variants whose kernels changed, the
# discriminator update alone at Ant width, the full bench line. Run through gpurun; copy the summaries to profiles/.
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03b; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
DB=$(find $O/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_bench.md | head -14
for v in P_mlp64_1024x16 P_ant_gail_d35; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$v -- python tools/variant_profile.py $v 8 > $O/kt_$v.log 2>&1
  DB=$(find $O/kt_$v -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$v.md | head -10
done
OD=27 AD=8 python tools/disc_step_bench.py 64 > $O/disc_step_wide_d35.txt 2>&1; tail -6 $O/disc_step_wide_d35.txt
python tools/ppo_epoch_timing.py > $O/ppo_epoch_timing.txt 2>&1; tail -3 $O/ppo_epoch_timing.txt | cut -c1-300
python tools/ppo_update_timing.py > $O/ppo_update_timing.txt 2>&1; tail -12 $O/ppo_update_timing.txt | cut -c1-200
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | head -40 || tail -c 600 $O/bench_full.json
find $O -name "*.db" -size +20M -delete; du -sh $O
