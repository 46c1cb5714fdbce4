#!/bin/bash
# Fourth gpurun call of the round-4 fault study: does the fault need the octet-major epilogue (v_permlane32_swap + 16-byte stores through a
# waterfall loop)?  BFSR_PAIR_DBG=fmt0 makes round 3's head write hid as plain NCHW dwords.  Output: gpurun_out/hf/log4.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $ROOT/gpurun_out/hf
LOG=$ROOT/gpurun_out/hf/log4.txt
: > $LOG
cd $ROOT/tools/exp/r3tree
run() {  # variant waves mode rounds
  echo "=== $1 waves=$2 $3 ($4 rounds) BFSR_PAIR_DBG=$BFSR_PAIR_DBG" >> $LOG
  BFSR_HIP_LIB=tools/exp/libhf_$1.so BFSR_HEAD_WAVES=$2 timeout 600 python tools/exp/head_fault.py $3 $4 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|UNet:" | tail -20 >> $LOG
}
run base 8 plain 300
BFSR_PAIR_DBG=fmt0 run base 8 plain 900
run base 8 plain 300
cat $LOG
