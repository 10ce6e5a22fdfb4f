#!/bin/bash
# SMALL form: rows staged by the idle waves -- bit-identity and same-box A/B through IA_LIB
for lib in imitation_amd/_ab/lib_stage0.so ""; do
  echo "lib=$lib"
  IA_LIB=$lib timeout 600 python tools/ppo_bits.py 2 3_airl_ant_tuned_verbatim 2>&1 | tail -2
  for i in 1 2; do IA_LIB=$lib timeout 600 python tools/ppo_step_us.py 0 6 3_airl_ant_tuned_verbatim 2>&1 | tail -1; done
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ppo_epochs_match_oracle or ppo_update" 2>&1 | tail -3
