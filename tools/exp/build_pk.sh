#!/bin/bash
# tools/exp/libpk.so = the library built WITH packed-fp32 VALU instructions (the compiler default; bfsr_amd/csrc/build.sh disables them, see NOPK there): for the
# same-box A/B of what the NOPK build costs (BFSR_HIP_LIB=$PWD/tools/exp/libpk.so python bench.py ...) and for tools/exp/victim_probe.py.  Not a product build.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/bfsr_amd/csrc"
mkdir -p build/pk
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
objs=""
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv_chain conv_up2_h2t conv_up4_h2t conv1x1 flow_ops coupling coupling_tail resample linf_ops linf_mlp metrics range_check; do
  $HIPCC $FLAGS -c $f.hip -o build/pk/$f.o &
  objs="$objs build/pk/$f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o "$R/tools/exp/libpk.so"
echo "built tools/exp/libpk.so"
