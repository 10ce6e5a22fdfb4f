#!/bin/bash
# same-box A/B of libraries built by tools/ab_libs.sh: bash tools/r06_ab.sh "<lib names>" "<variants ('P' = config P)>" [rounds]
O=gpurun_out/r06ab; mkdir -p $O
LIBS=${1:-"t32 t36"}; VARS=${2:-"P T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim"}; R=${3:-12}
for rep in 1 2; do
  for v in $VARS; do
    for l in $LIBS; do
      va=$v; [ "$v" = "P" ] && va=""
      echo -n "$l rep$rep: "; IA_LIB=imitation_amd/_ab/lib_$l.so timeout 600 python tools/ppo_step_us.py 0 $R $va 2>&1 | tail -1
    done
  done
done | tee $O/ab_$(echo $LIBS | tr ' ' '_').txt
