#!/bin/bash
# Measurement build of conv_chain_kernel with per-item phase timestamps (-DBFSR_CHAIN_TRACE=1 -> tools/exp/libchain_trace.so); run after
# bfsr_amd/csrc/build.sh, then on a GPU box: BFSR_HIP_LIB=$PWD/tools/exp/libchain_trace.so python tools/exp/chain_trace.py [B H NB]
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv_up2_h2t conv_up4_h2t conv1x1 flow_ops coupling coupling_tail coupling_wide resample linf_ops linf_mlp metrics range_check; do objs="$objs build/$f.o"; done
$HIPCC $FLAGS -DBFSR_CHAIN_TRACE=1 ${EXTRA:-} -c conv_chain.hip -o build/conv_chain_trace.o 2> >(grep -v "not a recognized feature for this target" >&2)
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/conv_chain_trace.o -o "$R/tools/exp/libchain_trace${SUFFIX:-}.so"
ls -la "$R"/tools/exp/libchain_trace*.so
