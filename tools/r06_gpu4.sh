#!/bin/bash
# hop-1 polls ahead of the gathers: parity of the PPO kernels under the new library, then the same-box A/B
O=gpurun_out/r06d; mkdir -p $O
IA_LIB=imitation_amd/_ab/lib_hop1.so timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ppo_epochs_match_oracle or spill or ppo_update" > $O/kern.log 2>&1; echo "kern rc=$?"; tail -3 $O/kern.log
bash tools/r06_ab.sh "hop0 hop1" "P 3_airl_ant_1024x16_mb1024 P_mlp64_1024x16" 12
IA_LIB=imitation_amd/_ab/lib_hop1.so timeout 600 python tools/ppo_update_timing.py 0 > $O/phase_P_hop1.txt 2>&1; grep "per step" $O/phase_P_hop1.txt | tail -1
IA_LIB=imitation_amd/_ab/lib_hop0.so timeout 600 python tools/ppo_update_timing.py 0 > $O/phase_P_hop0.txt 2>&1; grep "per step" $O/phase_P_hop0.txt | tail -1
