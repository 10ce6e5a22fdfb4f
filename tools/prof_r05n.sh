set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-200
python tools/ab_rounds.py T_gail_half_cheetah_tuned_verbatim gen.rollout_tail_one_call=True,False 150 3 2>&1 | grep ms/round | cut -c1-110 > $O/ab_tail.txt; cat $O/ab_tail.txt
