# Round 5: the whole GPU suite with ppo_epoch_ll2_kernel as the default 64-wide epoch kernel and the C-side replay-row predraw;
# 64-wide kernels A/B (round 4's body = mode 4), phase clocks, replay-row predraw A/B on whole rounds
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt | cut -c1-200
python tools/ppo_epoch_timing.py > $O/ll2_timing.txt 2>&1; tail -3 $O/ll2_timing.txt | cut -c1-420
for m in 4 0 4 0; do IA_EPOCH_SPLIT=$m timeout 300 python tools/variant_profile.py P_mlp64_1024x16 24 2>&1 | tail -1 | cut -c1-130; done > $O/mlp64_ab.txt; cat $O/mlp64_ab.txt
for m in 4 0 4 0; do IA_EPOCH_SPLIT=$m timeout 300 python tools/variant_profile.py 1_cartpole_8x256_mlp64 8 2>&1 | tail -1 | cut -c1-130; done > $O/cartpole_ab.txt; cat $O/cartpole_ab.txt
python tools/ab_rounds.py P predraw_disc_indices=True,False 150 3 2>&1 | grep ms/round > $O/P_predraw_ab.txt; cat $O/P_predraw_ab.txt
