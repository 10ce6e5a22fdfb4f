set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_adversarial_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "mailbox or golden or act" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-200
for V in P_gp10 P 1_cartpole_8x256_mlp64; do python tools/ab_rounds.py $V lib.ia_rollout_mailbox_sys_out=1,0 100 3 2>&1 | grep ms/round | cut -c1-100; done > $O/ab_sys.txt; cat $O/ab_sys.txt
for S in 1 0; do python - <<EOF
import sys; sys.path.insert(0, '.')
from imitation_amd import _lib as L
L.load().ia_rollout_mailbox_sys_out($S)
sys.argv = ['x', '20']
exec(open('tools/rollout_sections.py').read().split("# device time between")[0])
EOF
done 2>&1 | grep -E "us/step|sum" | cut -c1-120
