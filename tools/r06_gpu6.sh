#!/bin/bash
# AIRL + gradient penalty: merged launches. Bits old vs new, the penalty / AIRL tests, rounds A/B
O=gpurun_out/r06h; mkdir -p $O
for v in 3_airl_ant_1024x16_mb1024_gp10 3_airl_ant_1024x16_mb1024 P_gp10; do
  for l in vw2 new; do echo -n "$l: "; IA_LIB=imitation_amd/_ab/lib_$l.so timeout 300 python tools/ppo_bits.py 3 $v 2>&1 | tail -1; done
done | tee $O/bits.txt
timeout 1500 python -m pytest tests/test_grad_penalty_gpu.py tests/test_adversarial_gpu.py -m gpu -x -q -k "airl or penalty or gp" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for l in vw2 new; do for v in 3_airl_ant_1024x16_mb1024_gp10 3_airl_ant_1024x16_mb1024; do
  echo -n "$l: "; IA_LIB=imitation_amd/_ab/lib_$l.so timeout 300 python tools/variant_profile.py $v 40 2>&1 | tail -1 | cut -c1-120
done; done; done | tee $O/rounds.txt
