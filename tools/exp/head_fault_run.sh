#!/bin/bash
# One gpurun call of the round-4 fault study: confirm that this box reproduces the fault with the frozen 8-wave head, then run the
# differential builds (tools/exp/build_head_fault.sh) on the SAME box.  Output: gpurun_out/hf/log.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/hf
LOG=gpurun_out/hf/log.txt
: > $LOG
N=${HF_ROUNDS:-300}
run() {  # variant waves mode rounds
  echo "=== $1 waves=$2 $3 ($4 rounds)" >> $LOG
  BFSR_HIP_LIB=tools/exp/libhf_$1.so BFSR_HEAD_WAVES=$2 timeout 400 python tools/exp/head_fault.py $3 $4 2>&1 | grep -v "Warning\|warn" | tail -60 >> $LOG
}
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 >> $LOG
run base 8 plain $N
if grep -q "HF base waves=8 plain: 0 /" $LOG; then
  echo "box does not reproduce the fault: stopping" >> $LOG
  cat $LOG; exit 0
fi
run base 8 poison $N
run trace 8 trace $N
run voff 8 plain $N
run ldshigh 4 plain $N
run noprefetch 8 plain $N
run bar2 8 plain $N
run nop 8 plain $N
run sc1 8 plain $N
run forcezero 8 plain $N
run base 4 plain $N
run base 8 plain $N
cat $LOG
