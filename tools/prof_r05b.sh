# Round 5, image GAIL after the auxiliary-kernel fixes (avgpool by channel quads, grid-wide clip norm, implicit input gradient)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05b; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_general_policy_gpu.py tests/test_bc.py tests/test_adversarial_gpu.py -m gpu -x -q -k "not full_size" > $O/pytest_ops.txt 2>&1; tail -5 $O/pytest_ops.txt
rocprofv3 --kernel-trace --stats -d $O/kt_image -- python tools/variant_profile.py image_gail_64x16_cnn 3 > $O/kt_image.log 2>&1; tail -2 $O/kt_image.log | cut -c1-300
DB=$(find $O/kt_image -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_image_gail.md | head -24
python tools/variant_profile.py image_gail_64x16_cnn 6 2>&1 | tail -1 | cut -c1-200
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
