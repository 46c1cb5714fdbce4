#!/bin/bash
# round 6: deferred publish of the fused dense-block launch on / off (BFSR_CHAIN_DEFER), static deal.  gpurun: bash tools/exp/r06t.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r06t}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
export BFSR_CHAIN_DEAL=static
for d in 1 0; do
  echo "== trace build, BFSR_CHAIN_DEFER=$d, 8 160 69"; BFSR_CHAIN_DEFER=$d BFSR_HIP_LIB=$R/tools/exp/libchain_trace.so timeout 300 python tools/exp/chain_trace.py 8 160 69 2>&1 | grep -v amdgpu.ids
done > $OUT/${TAG}_chain_defer.txt 2>&1
cat $OUT/${TAG}_chain_defer.txt
(python tools/env_ab.py BFSR_CHAIN_DEFER 1 0 2>&1 | tail -4; python tools/env_ab.py BFSR_CHAIN_DEFER 1 0 --scale 8 --batch 64 --lr 96 2>&1 | tail -4; python tools/env_ab.py BFSR_CHAIN_DEFER 1 0 --scale 4 --batch 4 --lr 160 2>&1 | tail -4) > $OUT/${TAG}_ab_defer.txt 2>&1
cat $OUT/${TAG}_ab_defer.txt
