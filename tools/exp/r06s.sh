#!/bin/bash
# round 6: the fused dense-block launch with claimed items (one queue per XCD / one global queue) against the static deal (BFSR_CHAIN_DEAL=static|xcd|global).
# gpurun: bash tools/exp/r06s.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r06s}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_conv_chain.py -m gpu -x -q 2>&1 | tail -3) > $OUT/${TAG}_tests.txt
cat $OUT/${TAG}_tests.txt
for shape in "8 160 69" "64 96 24" "16 256 12" "16 96 24" "4 160 24"; do
  for d in static xcd global; do
    echo "== shipped build, BFSR_CHAIN_DEAL=$d, $shape"; BFSR_CHAIN_DEAL=$d timeout 300 python tools/exp/chain_trace.py $shape 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/${TAG}_chain_deal.txt 2>&1
for d in static xcd; do
  echo "== trace build, BFSR_CHAIN_DEAL=$d, 8 160 69"; BFSR_CHAIN_DEAL=$d BFSR_HIP_LIB=$R/tools/exp/libchain_trace.so timeout 300 python tools/exp/chain_trace.py 8 160 69 2>&1 | grep -v amdgpu.ids
done >> $OUT/${TAG}_chain_deal.txt 2>&1
cat $OUT/${TAG}_chain_deal.txt
(python tools/env_ab.py BFSR_CHAIN_DEAL static xcd 2>&1 | tail -4; python tools/env_ab.py BFSR_CHAIN_DEAL static xcd --scale 8 --batch 64 --lr 96 2>&1 | tail -4) > $OUT/${TAG}_ab_cfg2_cfg4.txt 2>&1
cat $OUT/${TAG}_ab_cfg2_cfg4.txt
