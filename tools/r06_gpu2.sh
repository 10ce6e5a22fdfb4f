#!/bin/bash
# phase clocks (measurement build) and production step times of the persistent PPO update, three shapes
O=gpurun_out/r06b
mkdir -p $O
for v in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim; do
  n=${v:-P}
  timeout 600 python tools/ppo_update_timing.py 0 $v > $O/phase_$n.txt 2>&1
  timeout 600 python tools/ppo_step_us.py 0 12 $v > $O/step_$n.txt 2>&1
  tail -12 $O/phase_$n.txt; tail -1 $O/step_$n.txt
done
