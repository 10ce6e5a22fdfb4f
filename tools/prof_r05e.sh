# Round 5: first run of the tower-resident 64-wide epoch kernel: parity, then speed against round 4's kernel
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "ppo_epochs_match_oracle or word_exchange" > $O/pytest_t64.txt 2>&1; tail -15 $O/pytest_t64.txt | cut -c1-250
for m in 4 0 4 0; do IA_EPOCH_SPLIT=$m timeout 300 python tools/variant_profile.py P_mlp64_1024x16 24 2>&1 | tail -1 | cut -c1-130; done > $O/mlp64_ab.txt; cat $O/mlp64_ab.txt
for m in 4 0; do IA_EPOCH_SPLIT=$m timeout 300 python tools/variant_profile.py 1_cartpole_8x256_mlp64 8 2>&1 | tail -1 | cut -c1-130; done > $O/cartpole_ab.txt; cat $O/cartpole_ab.txt
timeout 600 python -m pytest tests/test_distributed.py tests/test_adversarial_gpu.py -m gpu -x -q > $O/pytest_rest.txt 2>&1; tail -5 $O/pytest_rest.txt | cut -c1-250
python tools/ab_rounds.py P predraw_disc_indices=True,False 150 2 2>&1 | grep ms/round > $O/P_predraw_ab.txt; cat $O/P_predraw_ab.txt
