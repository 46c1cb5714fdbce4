#!/bin/bash
# Ablation builds of linf_mlp_kernel (tools/exp/libmlp_<mask>.so; BFSR_MLP_ABL bit 0 no weight loads, 1 no cf gathers, 2 no output stores, 3 no MFMAs (fp16 mode)).
# Run after bfsr_amd/csrc/build.sh; timed by tools/exp/mlp_abl_run.sh through BFSR_HIP_LIB (results are wrong by construction).
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv_chain conv_up2_h2t conv_up4_h2t conv1x1 flow_ops coupling coupling_tail resample linf_ops metrics range_check; do objs="$objs build/$f.o"; done
for n in ${MASKS:-1 2 4 8 3 7 15}; do
  ( $HIPCC $FLAGS -DBFSR_MLP_ABL=$n -c linf_mlp.hip -o build/linf_mlp_abl$n.o && $HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/linf_mlp_abl$n.o -o "$R/tools/exp/libmlp_$n.so" ) &
done
wait
ls "$R"/tools/exp/libmlp_*.so
