# Round 4, last refresh (after the 64-wide epoch kernel's second pass: ia_ppo_epochs, aligned b128 fragments): the full bench
# line, the kernel trace of the headline command and of P_mlp64_1024x16, the epoch kernels' phase clocks and A/Bs.
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04b; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
DB=$(find $O/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_bench.md | head -12
rocprofv3 --kernel-trace --stats -d $O/kt_mlp64 -- python tools/variant_profile.py P_mlp64_1024x16 6 > $O/kt_mlp64.log 2>&1
DB=$(find $O/kt_mlp64 -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_P_mlp64_1024x16.md | head -12
python tools/ppo_epoch_timing.py > $O/ppo_epoch_timing_ll.txt 2>&1; tail -5 $O/ppo_epoch_timing_ll.txt | cut -c1-500
IA_EPOCH_SPLIT=3 python tools/ppo_epoch_timing.py > $O/ppo_epoch_timing_barriers.txt 2>&1; tail -5 $O/ppo_epoch_timing_barriers.txt | cut -c1-500
for m in 3 0 3 0; do IA_EPOCH_SPLIT=$m python tools/variant_profile.py P_mlp64_1024x16 24 2>&1 | tail -1 | cut -c1-110; done > $O/mlp64_ab.txt; cat $O/mlp64_ab.txt
python tools/ab_rounds.py P_mlp64_1024x16 gen.epochs_one_call=True,False 60 2 2>&1 | grep ms/round > $O/mlp64_epochs_ab.txt; cat $O/mlp64_epochs_ab.txt
python tools/ab_rounds.py P disc_behind_ppo=None,True,False 100 2 2>&1 | grep ms/round > $O/P_schedule_ab.txt; cat $O/P_schedule_ab.txt
python tools/ab_rounds.py P disc_round_one_call=True,False 150 2 2>&1 | grep ms/round > $O/P_onecall_ab.txt; cat $O/P_onecall_ab.txt
python tools/dp_overhead.py 40 > $O/dp_overhead_sharded.txt 2>&1; head -8 $O/dp_overhead_sharded.txt
IA_DP_ROW_SHARDED=0 python tools/dp_overhead.py 40 > $O/dp_overhead_replicated.txt 2>&1; head -8 $O/dp_overhead_replicated.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | cut -c1-200 | head -24
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
