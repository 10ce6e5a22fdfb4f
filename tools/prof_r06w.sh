#!/bin/bash
# PMC passes (HBM traffic) of the image variant and the 64-wide variant: separate passes, --kernel-trace only beside --pmc
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06w; mkdir -p $O
for v in image_gail_64x16_cnn:3 P_mlp64_1024x16:6; do
  n=${v%%:*}; r=${v##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${n}_$c -- python tools/variant_profile.py $n $r > $O/pmc_${n}_$c.log 2>&1
    DB=$(find $O/pmc_${n}_$c -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB > $O/pmc_${n}_$c.txt
    grep -E "conv3x3|avgpool|conv1_fwd|ppo_epoch_ll2|reduce_bias" $O/pmc_${n}_$c.txt | cut -c1-70,92-
  done
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
