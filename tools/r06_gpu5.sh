#!/bin/bash
# value waves hand the normalised advantages / Gaussian constants over: bits, parity, same-box A/B
O=gpurun_out/r06f; mkdir -p $O
for v in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim 1_cartpole_8x256_mlp64; do
  for l in ${LIBS:-old vw}; do echo -n "$l: "; IA_LIB=imitation_amd/_ab/lib_$l.so timeout 300 python tools/ppo_bits.py 2 $v 2>&1 | tail -1; done
done | tee $O/bits.txt
IA_LIB=imitation_amd/_ab/lib_${NEW:-vw}.so timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ppo_epochs_match_oracle or spill or ppo_update" > $O/kern.log 2>&1; echo "kern rc=$?"; tail -3 $O/kern.log
bash tools/r06_ab.sh "${LIBS:-old vw}" "P T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim" 12
