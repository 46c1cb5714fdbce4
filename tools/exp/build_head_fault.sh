#!/bin/bash
# Differential builds for the round-4 study of the coupling_head wrong-half-tile fault (tools/exp/head_fault.py).
# tools/exp/libhf_<variant>.so = libbfsr_hip.so with build/coupling.o replaced by a build of tools/exp/kernels/coupling_r3.hip
# (round 3's kernel, frozen) under one switch.  Run after bfsr_amd/csrc/build.sh.  Selected at run time through BFSR_HIP_LIB.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
SRC="$R/tools/exp/kernels/coupling_r3.hip"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv1x1 flow_ops resample linf_ops linf_mlp metrics; do objs="$objs build/$f.o"; done
[ -f build/coupling_step.o ] && objs="$objs build/coupling_step.o"
mk() {  # name, extra flags...
  local n=$1; shift
  $HIPCC $FLAGS "$@" -c "$SRC" -o build/hf_$n.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/hf_$n.o -o "$R/tools/exp/libhf_$n.so"
}
mk base &
mk trace -DHF_TRACE &
mk nop -DHF_NOP &
mk noprefetch -DHF_NOPREFETCH &
mk noprez -DHF_NOPREZ &
mk noprep -DHF_NOPREP &
mk sc1 -DHF_SC1 &
mk bar2 -DHF_BAR2 &
mk bar2a -DHF_BAR2A &
mk bar2b -DHF_BAR2B &
mk zdb -DHF_ZDB &
mk ldshigh -DHF_LDSHIGH &
mk voff -DHF_VOFF &
mk forcezero -mllvm -amdgpu-waitcnt-forcezero=1 &
wait
ls -la "$R"/tools/exp/libhf_*.so
