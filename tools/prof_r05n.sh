set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.txt 2>&1; tail -14 $O/pytest.txt | cut -c1-200
for V in P P_gp10 P P_gp10; do python tools/ab_rounds.py $V predraw_round_draws=True 150 1 2>&1 | grep ms/round | cut -c1-90; done > $O/rounds.txt; cat $O/rounds.txt
python tools/round_timeline.py 10 1 > $O/round_timeline_P.txt 2>&1; sed -n 30,50p $O/round_timeline_P.txt | cut -c1-150
