#!/bin/bash
# AIRL round: what does not depend on the reward net runs ahead. Bits (ahead on/off, old library), tests, rounds
O=gpurun_out/r06i; mkdir -p $O
for v in 3_airl_ant_1024x16_mb1024_gp10 3_airl_ant_1024x16_mb1024 3_airl_ant_tuned_verbatim; do
  echo -n "vw2: "; IA_LIB=imitation_amd/_ab/lib_vw2.so IA_AIRL_ROUND_AHEAD=0 timeout 300 python tools/ppo_bits.py 3 $v 2>&1 | tail -1
  echo -n "new, per update: "; IA_AIRL_ROUND_AHEAD=0 timeout 300 python tools/ppo_bits.py 3 $v 2>&1 | tail -1
  echo -n "new, ahead: "; timeout 300 python tools/ppo_bits.py 3 $v 2>&1 | tail -1
done | tee $O/bits.txt
timeout 1500 python -m pytest tests/test_grad_penalty_gpu.py tests/test_adversarial_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q -k "airl or penalty or gp" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for ah in 0 1; do for v in 3_airl_ant_1024x16_mb1024_gp10 3_airl_ant_1024x16_mb1024 3_airl_ant_tuned_verbatim; do
  r=40; [ $v = 3_airl_ant_tuned_verbatim ] && r=6
  echo -n "ahead=$ah: "; IA_AIRL_ROUND_AHEAD=$ah timeout 300 python tools/variant_profile.py $v $r 2>&1 | tail -1 | cut -c1-120
done; done; done | tee $O/rounds.txt
