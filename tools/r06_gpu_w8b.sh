#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "two_waves_per_simd or word_exchange or (ppo_epochs_match_oracle and 64)" > gpurun_out/w8_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/w8_tests.log
tail -4 gpurun_out/w8_tests.log | cut -c1-200
for i in 1 2 3; do
  for m in 6 0; do
    echo "mode $m"; IA_EPOCH_SPLIT=$m timeout 600 python tools/variant_profile.py P_mlp64_1024x16 20 2>&1 | tail -1 | cut -c1-100
  done
done
IA_EPOCH_SPLIT=0 timeout 300 python tools/ppo_epoch_timing.py 2>&1 | tail -3
IA_EPOCH_SPLIT=6 timeout 300 python tools/ppo_epoch_timing.py 2>&1 | tail -3 | head -1
