# refresh of the final artefacts with the round's last library: kernel trace of the headline command, the full bench line
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05z2; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
DB=$(find $O/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_bench.md | head -10 | cut -c1-200
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | cut -c1-200 | head -30
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
