# wide-row penalty pass with eight column waves: parity, whole rounds
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 600 python -m pytest tests/test_disc_fused_gpu.py tests/test_grad_penalty_gpu.py -m gpu -q -x -k "penalty or gp" 2>&1 | tail -3 | cut -c1-200
python tools/ab_rounds.py P_ant_gail_d35_gp10 lib.ia_disc_fused_gp_groups=8,1 100 2 2>&1 | grep ms/round | cut -c1-110 > $O/ab_gp.txt; cat $O/ab_gp.txt
