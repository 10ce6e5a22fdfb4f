# one-launch penalty pass at H = 256 (eight column waves by default): full suite, whole rounds beside config P on the same box
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-200
for V in P P_gp10 P P_gp10; do python tools/ab_rounds.py $V predraw_round_draws=True 150 1 2>&1 | grep ms/round | cut -c1-90; done > $O/rounds.txt; cat $O/rounds.txt
