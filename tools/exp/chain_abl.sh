#!/bin/bash
# Ablation builds of conv_chain_kernel (tools/exp/libchain_<mask>.so; BFSR_CHAIN_ABL bit 0 no fragment reads, 1 no MFMAs, 2 no DMA, 3 no epilogue,
# 4 no store drain before the publish, 5 plain instead of sc1 loads / stores).  Run after bfsr_amd/csrc/build.sh; timed by tools/chain_bench.py
# through BFSR_HIP_LIB (results are wrong by construction).
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv_up2_h2t conv1x1 flow_ops coupling coupling_tail resample linf_ops linf_mlp metrics; do objs="$objs build/$f.o"; done
for n in ${MASKS:-1 2 3 4 5 8 16 32 13}; do
  ( $HIPCC $FLAGS -DBFSR_CHAIN_ABL=$n -c conv_chain.hip -o build/conv_chain_abl$n.o && $HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/conv_chain_abl$n.o -o "$R/tools/exp/libchain_$n.so" ) &
done
wait
ls "$R"/tools/exp/libchain_*.so
