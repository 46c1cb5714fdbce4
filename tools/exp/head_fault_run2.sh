#!/bin/bash
# Second gpurun call of the round-4 fault study: which barrier matters, does a double-buffered z1 tile remove the fault, where do the
# faulty rows sit, and a stand-alone barrier/LDS probe.  Output: gpurun_out/hf/log2.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/hf
LOG=gpurun_out/hf/log2.txt
: > $LOG
run() {  # variant waves mode rounds
  echo "=== $1 waves=$2 $3 ($4 rounds)" >> $LOG
  BFSR_HIP_LIB=tools/exp/libhf_$1.so BFSR_HEAD_WAVES=$2 timeout 600 python tools/exp/head_fault.py $3 $4 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|UNet:" | tail -80 >> $LOG
}
run base 8 plain 300
run zdb 8 plain 600
run bar2a 8 plain 600
run bar2b 8 plain 600
run base 8 check 600
echo "=== barrier_probe" >> $LOG
timeout 300 tools/exp/barrier_probe 20000 8 >> $LOG 2>&1
timeout 300 tools/exp/barrier_probe 20000 4 >> $LOG 2>&1
run noprefetch 8 plain 600
run bar2 8 plain 600
run base 8 plain 300
cat $LOG
