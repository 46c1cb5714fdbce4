#!/bin/bash
# round 6: per-item phase trace of the fused dense-block launch (config-2 shape and the config-4 shard), prior tile-threshold A/B.  gpurun: bash tools/exp/r06r.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06r; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for shape in "8 160 69" "64 96 24" "16 256 12"; do
  echo "== shipped build, $shape"; python tools/exp/chain_trace.py $shape 2>&1 | grep -v amdgpu.ids
  echo "== trace build, $shape"; BFSR_HIP_LIB=$R/tools/exp/libchain_trace.so python tools/exp/chain_trace.py $shape 2>&1 | grep -v amdgpu.ids
done > $OUT/${TAG}_chain_trace.txt 2>&1
(python tools/env_ab.py BFSR_PRIOR_MIN_TILES 32 16 --scale 8 --batch 64 --lr 96 2>&1 | tail -5; python tools/env_ab.py BFSR_PRIOR_MIN_TILES 32 16 --scale 8 --batch 8 --lr 96 2>&1 | tail -5; python tools/env_ab.py BFSR_PRIOR_MIN_TILES 32 12 2>&1 | tail -5) > $OUT/${TAG}_ab_prior_tiles.txt 2>&1
cat $OUT/${TAG}_chain_trace.txt $OUT/${TAG}_ab_prior_tiles.txt
