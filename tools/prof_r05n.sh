# fast_tanh with explicit fused multiply-adds: parity (all tests + the full-size deviations), then same-box step times
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-200
for L in base new; do
  LIBP=imitation_amd/libimitation_hip.so; [ $L = base ] && LIBP=imitation_amd/_ab/lib_base.so
  IA_LIB=$LIBP timeout 900 python -m pytest tests/test_adversarial_gpu.py -m gpu -q -s -k "full_size or horizon" 2>&1 | grep -E "worst deviation|passed|failed" | cut -c1-300 > $O/full_size_$L.txt; cat $O/full_size_$L.txt
done
for rep in 1 2 3; do
  for L in base new; do
    LIBP=imitation_amd/libimitation_hip.so; [ $L = base ] && LIBP=imitation_amd/_ab/lib_base.so
    for V in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim; do
      echo -n "$L: " >> $O/ppo_step_us.txt
      IA_LIB=$LIBP python tools/ppo_step_us.py 0 10 $V 2>/dev/null | tail -1 >> $O/ppo_step_us.txt
    done
    echo -n "$L: " >> $O/mlp64.txt; IA_LIB=$LIBP python tools/variant_profile.py P_mlp64_1024x16 24 2>&1 | tail -1 | cut -c1-110 >> $O/mlp64.txt
  done
done
cat $O/ppo_step_us.txt $O/mlp64.txt
