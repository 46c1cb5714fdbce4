#!/bin/bash
# One gpurun call: linf_mlp ablation builds timed inside the config-5 / config-3 pass (per-launch-shape table, the linf_mlp line).
cd "$(dirname "$0")/../.."
for c in 5 3; do
  echo "== config $c, full"; python tools/exp/linf_keys.py --config $c --top 12 2>&1 | grep linf_mlp
  for n in ${MASKS:-1 2 4 8 3 7 15}; do
    echo "== config $c BFSR_MLP_ABL=$n (bit 0 no weight loads, 1 no cf gathers, 2 no output stores, 3 no MFMAs (fp16 only))"
    BFSR_HIP_LIB=$PWD/tools/exp/libmlp_$n.so python tools/exp/linf_keys.py --config $c --top 12 2>&1 | grep linf_mlp
  done
done
