set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05f; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "ppo_epochs_match_oracle or word_exchange" > $O/pytest_t64.txt 2>&1; tail -4 $O/pytest_t64.txt | cut -c1-250
python tools/ppo_epoch_timing.py > $O/t64_timing.txt 2>&1; tail -3 $O/t64_timing.txt | cut -c1-400
python tools/ppo_epoch_timing.py 1_cartpole_8x256_mlp64 > $O/t64_timing_cartpole.txt 2>&1; tail -3 $O/t64_timing_cartpole.txt | cut -c1-400
for m in 4 0 4 0; do IA_EPOCH_SPLIT=$m timeout 300 python tools/variant_profile.py P_mlp64_1024x16 24 2>&1 | tail -1 | cut -c1-130; done > $O/mlp64_ab.txt; cat $O/mlp64_ab.txt
