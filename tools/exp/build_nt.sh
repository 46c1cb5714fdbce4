#!/bin/bash
# Measurement build: the large streamed outputs of the hoisted conditioning convs stored with another cache policy (-DBFSR_OUT_AUX=<aux>: 2 = nt, 16 = sc1, 18 = both)
# -> tools/exp/libchain_out<aux>.so (the libchain_ prefix keeps it git-ignored).  Run after bfsr_amd/csrc/build.sh.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops"
for aux in ${AUXES:-2}; do
  objs=""
  for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_chain conv1x1 flow_ops coupling coupling_tail coupling_wide resample linf_ops linf_mlp metrics range_check; do objs="$objs build/$f.o"; done
  for f in conv_h2s conv_up2_h2t conv_up4_h2t; do
    $HIPCC $FLAGS -DBFSR_OUT_AUX=$aux -c $f.hip -o build/${f}_out$aux.o 2> >(grep -v "not a recognized feature for this target" >&2) &
    objs="$objs build/${f}_out$aux.o"
  done
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o "$R/tools/exp/libchain_out$aux.so"
done
ls -la "$R"/tools/exp/libchain_out*.so
