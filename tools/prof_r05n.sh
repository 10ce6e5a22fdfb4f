set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
for R in 64 0 32 0 64; do IA_DISC_RESERVE_CUS=$R python tools/image_timing.py 2>&1 | tail -1 | cut -c1-400; done > $O/image.txt; cat $O/image.txt
