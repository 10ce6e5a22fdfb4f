# module nets / image policy in the overlapped schedules, act-step wave priority, round draws: parity, then same-box A/Bs
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt | cut -c1-300
for i in 1 2; do python tools/variant_profile.py image_gail_64x16_cnn 3 2>&1 | tail -1 | cut -c1-120; done > $O/image.txt; cat $O/image.txt
for V in P_gp10 P; do
  python tools/ab_rounds.py $V lib.ia_rollout_mailbox_prio=1,0 100 3 2>&1 | grep ms/round >> $O/ab_prio.txt
done
cat $O/ab_prio.txt
python tools/host_profile.py image_gail_64x16_cnn 3 2>&1 | grep -v "^$" | cut -c1-150 | sed "s#/tmp/code/[^ ]*/repo/##" | head -50 > $O/host_profile_image.txt
