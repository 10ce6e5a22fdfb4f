#!/bin/bash
# SQ instruction / stall counters of the persistent PPO update alone (the update of `tools/ppo_step_us.py`), separate passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06pmc; mkdir -p $O
V=${1:-}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CU_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -- python tools/ppo_step_us.py 0 4 $V > $O/p$i.log 2>&1
  DB=$(find $O/p$i -name "*results.db" | head -1); python tools/rocpd_pmc.py $DB | grep -E "ppo_update_persistent" | cut -c1-70,92- > $O/pmc$i${V:+_$V}.txt; cat $O/pmc$i${V:+_$V}.txt
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
