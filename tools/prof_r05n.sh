set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_disc_fused_gpu.py -m gpu -q -x -s -k "fused_prediction" 2>&1 | grep -E "fused prediction|passed|failed|Error|assert" | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-200
python tools/tail_timeline.py 60 2>&1 | tail -8
for F in 1 0 1 0; do IA_FUSED_PREDICT=$F python tools/ab_rounds.py P predraw_round_draws=True 150 1 2>&1 | grep ms/round | cut -c1-80; done
