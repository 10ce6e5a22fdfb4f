#!/bin/bash
# Round-6 baseline kernel traces of the variants the round works on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06g; mkdir -p $O
for v in ${VARS:-3_airl_ant_1024x16_mb1024_gp10:8 3_airl_ant_1024x16_mb1024:8 image_gail_64x16_cnn:3 P_mlp64_1024x16:8 P_gp10:8}; do
  n=${v%%:*}; r=${v##*:}
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/variant_profile.py $n $r > $O/kt_$n.log 2>&1
  DB=$(find $O/kt_$n -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$n.md | head -${HEAD:-22}
  tail -1 $O/kt_$n.log | cut -c1-200
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
