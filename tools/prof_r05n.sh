# the rollout's tail as one host call: parity, host timeline
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-200
python tools/tail_timeline.py 60 2>&1 | tail -7
