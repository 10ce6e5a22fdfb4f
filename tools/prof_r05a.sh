# Round 5, first call: the GPU suite (incl. the SB3 bookkeeping pin on the product), the generic SB3-protocol VecEnv at 1 024
# envs (rounds + host timeline), image GAIL (kernel trace + host profile), per-step baselines, the full bench line.
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python tools/variant_profile.py P_generic_vecenv_1024 24 2>&1 | tail -1 | cut -c1-200 > $O/generic_1024.txt; cat $O/generic_1024.txt
python tools/round_timeline.py 12 1 P_generic_vecenv_1024 > $O/generic_1024_timeline.txt 2>&1; tail -40 $O/generic_1024_timeline.txt
python tools/host_profile.py P_generic_vecenv_1024 10 > $O/generic_1024_host_profile.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_image -- python tools/variant_profile.py image_gail_64x16_cnn 3 > $O/kt_image.log 2>&1
DB=$(find $O/kt_image -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_image_gail.md | head -30
python tools/host_profile.py image_gail_64x16_cnn 3 > $O/image_host_profile.txt 2>&1; head -60 $O/image_host_profile.txt | cut -c1-180
python tools/ppo_step_us.py 0 12 > $O/ppo_step_us.txt 2>&1; python tools/ppo_step_us.py 0 12 P_mlp64_1024x16 >> $O/ppo_step_us.txt 2>&1; cat $O/ppo_step_us.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | cut -c1-200 | head -40
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
