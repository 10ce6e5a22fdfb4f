#!/bin/bash
# hardware-queue aliasing of the side streams? rounds with 4 (default) and 8 hardware queues
O=gpurun_out/r06j; mkdir -p $O
for rep in 1 2; do for q in 4 8; do for v in 3_airl_ant_1024x16_mb1024_gp10 3_airl_ant_1024x16_mb1024 P_gp10 P_mlp64_1024x16; do
  echo -n "queues=$q: "; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/variant_profile.py $v 40 2>&1 | tail -1 | cut -c1-120
done; done; done | tee $O/queues.txt
for q in 4 8; do echo "queues=$q"; GPU_MAX_HW_QUEUES=$q python tools/ab_rounds.py P disc_behind_ppo=None 100 2 2>&1 | tail -2; done | tee -a $O/queues.txt
