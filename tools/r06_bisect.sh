#!/bin/bash
# which test file, run ahead of it, makes the first packed `update` case of test_ppo_epochs_match_oracle abort?
O=gpurun_out/r06n; mkdir -p $O
K='test_ppo_epochs_match_oracle and update-11-3-32'
for f in test_adversarial_gpu test_checkpoint_gpu test_conv3x3_gpu test_disc_fused_gpu test_distributed test_grad_penalty_gpu test_host_logic; do
  timeout 900 python -m pytest tests/$f.py tests/test_kernels_gpu.py -m gpu -x -q -k "($f) or ($K)" -p no:cacheprovider > $O/$f.txt 2>&1
  echo "$f + case: rc=$? $(tail -1 $O/$f.txt | cut -c1-100)"
done
