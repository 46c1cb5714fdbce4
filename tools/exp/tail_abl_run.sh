#!/bin/bash
# coupling tail: product vs ablation builds (tools/exp/tail_abl.sh), level-1 and level-2 shapes of BASELINE config 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in product 1 2 3; do
  echo "=== tail build: $n   (1 = no epilogue, 2 = no LDS reads / MFMAs, 3 = DMA + barriers only)"
  if [ $n = product ]; then L=bfsr_amd/lib/libbfsr_hip.so; else L=tools/exp/libtail_$n.so; fi
  BFSR_HIP_LIB=$L python tools/step_bench.py 2 2>&1 | grep "quad-major=1" | grep "320x320\|160x160" | sed 's/head [^t]*//'
done
