#!/bin/bash
# One gpurun call: the conv_chain ablation builds timed on the dense-block convs (chain-of-one) at 8 x 160^2 and 16 x 256^2.
cd "$(dirname "$0")/../.."
for shape in "8 160" "16 256"; do
  echo "== full, $shape"; python tools/exp/chain_one.py $shape 2>&1 | grep -v amdgpu.ids
  for n in ${MASKS:-1 2 3 4 5 8 16 32 13}; do
    echo "== BFSR_CHAIN_ABL=$n (bit 0 no fragment reads, 1 no MFMAs, 2 no DMA, 3 no epilogue, 4 no drain, 5 plain loads/stores), $shape"
    BFSR_HIP_LIB=$PWD/tools/exp/libchain_$n.so python tools/exp/chain_one.py $shape 2>&1 | grep -v amdgpu.ids
  done
done
