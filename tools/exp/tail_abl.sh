#!/bin/bash
# Ablation builds of coupling_tail.hip (tools/exp/libtail_<n>.so): 1 = no epilogue, 2 = no LDS reads / MFMAs (barrier sequence and DMA kept),
# 3 = both (DMA + barriers only).  Run after bfsr_amd/csrc/build.sh; timed by tools/step_bench.py through BFSR_HIP_LIB.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv1x1 flow_ops coupling resample linf_ops linf_mlp metrics; do objs="$objs build/$f.o"; done
for n in 1 2 3; do
  $HIPCC $FLAGS -DBFSR_TAIL_ABL=$n -c coupling_tail.hip -o build/coupling_tail_abl$n.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/coupling_tail_abl$n.o -o "$R/tools/exp/libtail_$n.so"
done
ls -la "$R"/tools/exp/libtail_*.so
