#!/bin/bash
# Build libbfsr_hip.so for gfx950 (cross-compiles without a GPU).  Usage: bfsr_amd/csrc/build.sh [--clean]
set -e
cd "$(dirname "$0")"
OUT=../lib
if [ "$1" = "--clean" ]; then rm -rf build "$OUT"/libbfsr_hip.so; fi
mkdir -p "$OUT" build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# NOPK: no packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in device code.  Measured on MI355X (round 5; tools/exp/sgpr_war_probe.py,
# tools/exp/victim_probe.py): a dependent chain of them returns wrong values in a few hundred lanes per launch when the wave shares its SIMD with MFMA waves of
# ANOTHER kernel (two streams) -- hipcc -O3 forms such chains by SLP vectorisation in plain kernels (the bilinear resize went wrong under the prior / flow overlap).
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $NOPK"
pids=()
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv_chain conv_up2_h2t conv_up4_h2t conv1x1 flow_ops coupling coupling_tail coupling_wide resample linf_ops linf_mlp metrics range_check; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ ../../include/bfsr_hip.h -nt build/$f.o ] || [ launch_util.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o 2> >(grep -v "not a recognized feature for this target" >&2) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/conv_mfma.o build/conv_f16.o build/conv_bf16x3.o build/conv_x3s.o build/conv_h2s.o build/conv_chain.o build/conv_up2_h2t.o build/conv_up4_h2t.o build/conv1x1.o build/flow_ops.o build/coupling.o build/coupling_tail.o build/coupling_wide.o build/resample.o build/linf_ops.o build/linf_mlp.o build/metrics.o build/range_check.o -o "$OUT/libbfsr_hip.so"
echo "built $OUT/libbfsr_hip.so"
