#!/bin/bash
# Third gpurun call of the round-4 fault study: which of the two register prefetches matters (z1 under the 3x3, pre_aff under the 1x1 and
# the stores), and the stand-alone MFMA + in-flight-loads probe.  Runs inside tools/exp/r3tree (a worktree of the round-3 commit, since
# the product's coupling ABI changed in round 4).  Output: gpurun_out/hf/log3.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $ROOT/gpurun_out/hf
LOG=$ROOT/gpurun_out/hf/log3.txt
: > $LOG
echo "=== mfma_vmem_probe" >> $LOG
for w in 8 4; do for l in 1 0; do timeout 300 $ROOT/tools/exp/mfma_vmem_probe 30000 $w $l >> $LOG 2>&1; done; done
cd $ROOT/tools/exp/r3tree
run() {  # variant waves mode rounds
  echo "=== $1 waves=$2 $3 ($4 rounds)" >> $LOG
  BFSR_HIP_LIB=tools/exp/libhf_$1.so BFSR_HEAD_WAVES=$2 timeout 600 python tools/exp/head_fault.py $3 $4 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|UNet:" | tail -20 >> $LOG
}
run base 8 plain 300
run noprez 8 plain 600
run noprep 8 plain 600
BFSR_PAIR_DBG=fmt0 run base 8 plain 600
run base 8 plain 300
cat $LOG
