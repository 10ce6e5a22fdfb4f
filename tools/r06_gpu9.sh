#!/bin/bash
O=gpurun_out/r06k; mkdir -p $O
for rep in 1 2; do for ah in 0 1; do for v in 3_airl_ant_1024x16_mb1024_gp10 3_airl_ant_1024x16_mb1024 3_airl_ant_tuned_verbatim P_gp10; do
  r=40; [ $v = 3_airl_ant_tuned_verbatim ] && r=6
  echo -n "ahead=$ah: "; IA_AIRL_ROUND_AHEAD=$ah timeout 300 python tools/variant_profile.py $v $r 2>&1 | tail -1 | cut -c1-120
done; done; done | tee $O/rounds.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
