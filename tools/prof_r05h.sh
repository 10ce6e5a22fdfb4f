# Round 5: distributed tests after the data-parallel changes; the penalty variant's kernel trace + host timeline; image GAIL
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q > $O/pytest_dist.txt 2>&1; tail -3 $O/pytest_dist.txt | cut -c1-200
rocprofv3 --kernel-trace --stats -d $O/kt_gp -- python tools/variant_profile.py P_gp10 12 > $O/kt_gp.log 2>&1; tail -1 $O/kt_gp.log | cut -c1-200
DB=$(find $O/kt_gp -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_P_gp10.md > /dev/null; head -14 $O/kernel_stats_P_gp10.md | cut -c1-170
python tools/round_timeline.py 8 1 P_gp10 > $O/P_gp10_timeline.txt 2>&1; tail -24 $O/P_gp10_timeline.txt
python tools/host_profile.py P_gp10 10 > $O/P_gp10_host_profile.txt 2>&1; head -40 $O/P_gp10_host_profile.txt | cut -c1-150
python tools/variant_profile.py image_gail_64x16_cnn 6 2>&1 | tail -1 | cut -c1-200
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
