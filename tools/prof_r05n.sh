set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_disc_fused_gpu.py tests/test_grad_penalty_gpu.py tests/test_adversarial_gpu.py -m gpu -q -x -k "penalty or gp or one_call or round_draws" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-200
for V in P_gp10 P_ant_gail_d35_gp10; do python tools/ab_rounds.py $V lib.ia_disc_fused_tn_pair=1,0 100 3 2>&1 | grep ms/round | cut -c1-100; done > $O/ab_pair.txt; cat $O/ab_pair.txt
rocprofv3 --kernel-trace --stats -d $O/kt_gp -- python tools/variant_profile.py P_gp10 6 > $O/kt_gp.log 2>&1
DB=$(find $O/kt_gp -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_P_gp10.md | head -12 | cut -c1-200
find $O -name "*.db" -delete
