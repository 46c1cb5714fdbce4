#!/bin/bash
# One gpurun call: the conv_h2x ablation builds timed on the RDB shapes of config 2 (8 x 160^2).
cd "$(dirname "$0")/../.."
echo "== full"; python tools/exp/h2x_bench.py 8 160 160 f16x2 2>&1 | grep -v amdgpu.ids
for n in ${MASKS:-1 2 3 4 8 12 13 14}; do
  echo "== BFSR_H2X_ABL=$n (bit 0 no fragment reads, 1 no MFMAs, 2 no DMA, 3 no epilogue)"
  BFSR_HIP_LIB=$PWD/tools/exp/libh2x_$n.so python tools/exp/h2x_bench.py 8 160 160 f16x2 2>&1 | grep -v amdgpu.ids
done
