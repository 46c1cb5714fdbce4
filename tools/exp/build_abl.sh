#!/bin/bash
# tools/exp/libabl.so = libbfsr_hip.so with the ablation switches of conv_x3s / conv_h2s compiled in (-DBFSR_X3S_ABL -DBFSR_H2S_ABL);
# used by tools/exp/x3s_abl.py and tools/exp/h2s_abl.sh through BFSR_HIP_LIB.  Run after bfsr_amd/csrc/build.sh.
set -e
cd "$(dirname "$0")/../../bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
$HIPCC $FLAGS -DBFSR_X3S_ABL -I. -c ../../tools/exp/kernels/conv_x3s_r3.hip -o build/conv_x3s_abl.o &
$HIPCC $FLAGS -DBFSR_H2S_ABL -I. -c ../../tools/exp/kernels/conv_h2s_r3.hip -o build/conv_h2s_abl.o &
wait
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv1x1 flow_ops coupling coupling_step resample linf_ops linf_mlp metrics; do objs="$objs build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/conv_x3s_abl.o build/conv_h2s_abl.o -o ../../tools/exp/libabl.so
echo "built tools/exp/libabl.so"
