#!/bin/bash
# A/B of the padded transposed image: PPO kernel tests + step times for three shapes
O=gpurun_out/r06c
mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ppo_epochs_match_oracle or spill" > $O/kern.log 2>&1; echo "kern rc=$?"; tail -3 $O/kern.log
for v in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim 3_airl_ant_1024x16_mb1024; do
  n=${v:-P}
  timeout 600 python tools/ppo_step_us.py 0 12 $v > $O/step_$n.txt 2>&1
  tail -1 $O/step_$n.txt
done
timeout 600 python tools/ppo_update_timing.py 0 > $O/phase_P.txt 2>&1; grep "per step" $O/phase_P.txt | tail -1
