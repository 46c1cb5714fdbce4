#!/bin/bash
# Ablation builds of conv3x3_h2x_kernel (tools/exp/libh2x_<mask>.so; BFSR_H2X_ABL bit 0 no fragment reads, 1 no MFMAs, 2 no DMA, 3 no epilogue).
# Run after bfsr_amd/csrc/build.sh; timed by tools/exp/h2x_bench.py through BFSR_HIP_LIB (results are wrong by construction).
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv1x1 flow_ops coupling coupling_tail resample linf_ops linf_mlp metrics; do objs="$objs build/$f.o"; done
for n in ${MASKS:-1 2 3 4 8 12 13 14}; do
  ( $HIPCC $FLAGS -DBFSR_H2X_ABL=$n -c conv_h2s.hip -o build/conv_h2s_abl$n.o && $HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/conv_h2s_abl$n.o -o "$R/tools/exp/libh2x_$n.so" ) &
done
wait
ls "$R"/tools/exp/libh2x_*.so
