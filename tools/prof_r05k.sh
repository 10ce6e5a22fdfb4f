# the VALU-dot tiles back for the multi-workgroup forms: parity, then same-box step times
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05k; mkdir -p $O; rm -f $O/*.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_distributed.py -m gpu -q -k "ppo_epochs_match_oracle or world_two" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-200
for rep in 1 2 3; do
  for L in r04policy all3 final; do
    LIBP=imitation_amd/_ab/lib_$L.so; [ $L = final ] && LIBP=imitation_amd/libimitation_hip.so
    for V in "" 3_airl_ant_1024x16_mb1024 T_gail_half_cheetah_tuned_verbatim; do
      echo -n "$L: " >> $O/ppo_step_us.txt
      IA_LIB=$LIBP python tools/ppo_step_us.py 0 10 $V 2>/dev/null | tail -1 >> $O/ppo_step_us.txt
    done
  done
done
cat $O/ppo_step_us.txt
for rep in 1 2 3; do for L in r04policy final; do
  LIBP=imitation_amd/_ab/lib_$L.so; [ $L = final ] && LIBP=imitation_amd/libimitation_hip.so
  echo -n "$L: " >> $O/rounds.txt; IA_LIB=$LIBP python tools/ab_rounds.py P pipeline_rounds=True 200 1 2>&1 | grep ms/round >> $O/rounds.txt
done; done; cat $O/rounds.txt
