# Round-5 baseline measurement set (after the re-entry): suite, headline kernel trace + PMC, variant traces + host profiles
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt | cut -c1-200
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
DB=$(find $O/kt -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_bench.md | head -16
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python bench.py --steps 8 --warmup 3 --no-variants --no-cpu-baseline --prof-rounds 0 > /dev/null 2> $O/pmc_$c.log
  DB=$(find $O/pmc_$c -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB > $O/pmc_$c.txt; grep -E "ppo_update_persistent|disc_fb|disc_reduce|ia_gemm_tn_side|disc_assemble|rn_merge_seq" $O/pmc_$c.txt | cut -c1-60,92-
done
for v in image_gail_64x16_cnn:3 P_gp10:6; do
  n=${v%%:*}; r=${v##*:}
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/variant_profile.py $n $r > $O/kt_$n.log 2>&1
  DB=$(find $O/kt_$n -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$n.md | head -14
done
python tools/ppo_update_timing.py 0 > $O/ppo_timing_P.txt 2>&1; tail -14 $O/ppo_timing_P.txt | cut -c1-400
python tools/host_profile.py P_gp10 10 2>&1 | grep -v "^$" | cut -c1-150 | sed "s#/tmp/code/[^ ]*/repo/##" | head -60 > $O/host_profile_gp10.txt
python tools/host_profile.py image_gail_64x16_cnn 3 2>&1 | grep -v "^$" | cut -c1-150 | sed "s#/tmp/code/[^ ]*/repo/##" | head -60 > $O/host_profile_image.txt
python tools/host_profile.py P_generic_vecenv_1024 10 2>&1 | grep -v "^$" | cut -c1-150 | sed "s#/tmp/code/[^ ]*/repo/##" | head -60 > $O/host_profile_generic.txt
python tools/round_timeline.py 8 1 > $O/round_timeline_P.txt 2>&1
for v in P_stagger_arrays_1024 P_generic_vecenv_1024 P_stagger_arrays_1024 P_generic_vecenv_1024 P P_gp10 P P_gp10; do python tools/variant_profile.py $v 24 2>&1 | tail -1 | cut -c1-110; done > $O/pairs.txt; cat $O/pairs.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.log; python tools/show_bench.py $O/bench_full.json 2>/dev/null | cut -c1-300 | head -40
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
