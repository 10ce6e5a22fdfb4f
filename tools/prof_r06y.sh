#!/bin/bash
# Round-6 closing traces: kernel stats of the image variant and the 64-wide variant with the final code (one gpurun call)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06y; mkdir -p $O
for v in image_gail_64x16_cnn:3 P_mlp64_1024x16:6; do
  n=${v%%:*}; r=${v##*:}
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/variant_profile.py $n $r > $O/kt_$n.log 2>&1
  DB=$(find $O/kt_$n -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$n.md | head -14 | cut -c1-200
  tail -1 $O/kt_$n.log | cut -c1-160
done
python tools/host_profile.py image_gail_64x16_cnn 3 2>&1 | tail -40 | cut -c1-200 > $O/host_profile_image.txt; head -30 $O/host_profile_image.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
