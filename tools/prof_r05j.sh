# Round 5: same-box A/B of the final library against round 4's policy.hip (built into imitation_amd/_ab/lib_r04policy.so)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05j; mkdir -p $O
for rep in 1 2 3; do
  for L in r04policy final; do
    LIBP=imitation_amd/_ab/lib_$L.so; [ $L = final ] && LIBP=imitation_amd/libimitation_hip.so
    for V in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim 3_airl_ant_1024x16_mb1024; do
      echo -n "$L: " >> $O/ppo_step_us.txt
      IA_LIB=$LIBP python tools/ppo_step_us.py 0 10 $V 2>/dev/null | tail -1 >> $O/ppo_step_us.txt
    done
  done
done
cat $O/ppo_step_us.txt
for rep in 1 2; do for L in r04policy final; do
  LIBP=imitation_amd/_ab/lib_$L.so; [ $L = final ] && LIBP=imitation_amd/libimitation_hip.so
  echo -n "$L: " >> $O/rounds.txt; IA_LIB=$LIBP python tools/ab_rounds.py P pipeline_rounds=True 200 1 2>&1 | grep ms/round >> $O/rounds.txt
done; done; cat $O/rounds.txt
for v in P_stagger_arrays_1024 P_generic_vecenv_1024 P_stagger_arrays_1024 P_generic_vecenv_1024; do python tools/variant_profile.py $v 24 2>&1 | tail -1 | cut -c1-110; done > $O/generic_pair.txt; cat $O/generic_pair.txt
