#!/bin/bash
O=gpurun_out/r06n; mkdir -p $O
K='test_ppo_epochs_match_oracle and update-11-3-32'
ps aux | grep -c python
AMD_LOG_LEVEL=1 timeout 900 python -m pytest tests/test_distributed.py tests/test_kernels_gpu.py -m gpu -x -q -s -k "(test_distributed) or ($K)" -p no:cacheprovider > $O/pair_log.txt 2>&1; echo "rc=$?"
grep -v "^  File\|^Thread\|Extension modules" $O/pair_log.txt | grep -i -B2 -A6 "error\|fault\|abort\|hsa_status" | head -60 | cut -c1-250
ps aux | grep python | grep -v grep | cut -c1-150 | head
nvidia-smi 2>/dev/null; rocm-smi --showpids 2>/dev/null | head -20
dmesg 2>/dev/null | tail -5
