#!/bin/bash
# tools/exp/libstep.so = libbfsr_hip.so with coupling_step.hip's phase trace compiled in (-DBFSR_STEP_TRACE); used by
# tools/exp/step_trace.py through bfsr_amd._lib.LIB_PATH.  Run after bfsr_amd/csrc/build.sh.
set -e
cd "$(dirname "$0")/../../bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv1x1 flow_ops coupling resample linf_ops linf_mlp metrics; do objs="$objs build/$f.o"; done
# usage: build_step_trace.sh [V ...]: one library per experiment mask V (default 0) -> tools/exp/libstep[_V].so
for V in ${@:-0}; do
  $HIPCC $FLAGS -DBFSR_STEP_TRACE -DBFSR_STEP_V=$V -c coupling_step.hip -o build/coupling_step_trace_$V.o
  out=../../tools/exp/libstep.so; [ "$V" != "0" ] && out=../../tools/exp/libstep_$V.so
  $HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/coupling_step_trace_$V.o -o $out
  echo "built $out"
done
