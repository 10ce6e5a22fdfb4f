# Round 5: same-box A/B of the persistent PPO update's changes (one library per switch, tools/ab_libs.sh), parity first.
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_general_policy_gpu.py -m gpu -x -q -k "ppo or clip or stage or persistent or update" > $O/pytest_ppo_all.txt 2>&1; tail -3 $O/pytest_ppo_all.txt
for L in pf16 adam tiles; do
  IA_LIB=imitation_amd/_ab/lib_$L.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ppo_epochs_match_oracle" > $O/pytest_ppo_$L.txt 2>&1; tail -2 $O/pytest_ppo_$L.txt
done
for rep in 1 2; do
  for L in base pf16 adam tiles all; do
    LIBP=imitation_amd/_ab/lib_$L.so; [ $L = all ] && LIBP=imitation_amd/libimitation_hip.so
    for V in "" T_gail_half_cheetah_tuned_verbatim 3_airl_ant_tuned_verbatim 3_airl_ant_1024x16_mb1024; do
      echo -n "$L: " >> $O/ppo_step_us.txt
      IA_LIB=$LIBP python tools/ppo_step_us.py 0 10 $V 2>/dev/null | tail -1 >> $O/ppo_step_us.txt
    done
  done
done
cat $O/ppo_step_us.txt
for L in base all; do
  LIBP=imitation_amd/_ab/lib_$L.so; [ $L = all ] && LIBP=imitation_amd/libimitation_hip.so
  IA_LIB=$LIBP timeout 900 python -m pytest tests/test_adversarial_gpu.py -m gpu -q -s -k "full_size or horizon" 2>&1 | grep -E "worst deviation|passed|failed" > $O/full_size_$L.txt; cat $O/full_size_$L.txt
done
python tools/variant_profile.py image_gail_64x16_cnn 6 2>&1 | tail -1 | cut -c1-200 > $O/image.txt; cat $O/image.txt
