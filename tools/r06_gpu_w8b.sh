#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "two_waves_per_simd or word_exchange or (ppo_epochs_match_oracle and 64)" > gpurun_out/w8_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/w8_tests.log
tail -12 gpurun_out/w8_tests.log | cut -c1-250
python tools/ab_rounds.py P_mlp64_1024x16 lib.ia_ppo_epoch_split=7,0 40 4 2>&1 | grep "ms/round" | cut -c1-110
IA_EPOCH_SPLIT=0 timeout 300 python tools/ppo_epoch_timing.py 2>&1 | tail -3
