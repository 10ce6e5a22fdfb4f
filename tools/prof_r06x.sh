#!/bin/bash
# kernel trace of the general-towers variant (generic MFMA stacks, hipGraph-replayed update)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06x; mkdir -p $O
n=towers_1024x16_pi128x64_vf256
rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/variant_profile.py $n 6 > $O/kt_$n.log 2>&1
DB=$(find $O/kt_$n -name "*results.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$n.md | head -24 | cut -c1-200
tail -1 $O/kt_$n.log | cut -c1-160
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
