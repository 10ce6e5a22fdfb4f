# Round 5: GPU suite with the index predraw + final PPO kernel; headline A/B of the predraw; bench headline; image profile
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python tools/ab_rounds.py P predraw_disc_indices=True,False 150 2 2>&1 | grep ms/round > $O/P_predraw_ab.txt; cat $O/P_predraw_ab.txt
python tools/ab_rounds.py P_generic_vecenv_1024 predraw_disc_indices=True,False 60 2 2>&1 | grep ms/round > $O/generic_predraw_ab.txt; cat $O/generic_predraw_ab.txt
python tools/host_profile.py P 20 > $O/P_host_profile.txt 2>&1; head -50 $O/P_host_profile.txt | cut -c1-150
python tools/round_timeline.py 8 > $O/P_timeline.txt 2>&1; tail -30 $O/P_timeline.txt
python bench.py --no-variants > $O/bench_headline.json 2> $O/bench_headline.log; python tools/show_bench.py $O/bench_headline.json 2>/dev/null | cut -c1-200 | head -8
rocprofv3 --kernel-trace --stats -d $O/kt_image -- python tools/variant_profile.py image_gail_64x16_cnn 3 > $O/kt_image.log 2>&1
DB=$(find $O/kt_image -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_image_gail.md > /dev/null; head -16 $O/kernel_stats_image_gail.md | cut -c1-150
python tools/host_profile.py image_gail_64x16_cnn 3 > $O/image_host_profile.txt 2>&1; head -40 $O/image_host_profile.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
