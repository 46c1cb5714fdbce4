#!/bin/bash
# Build libbfsr_hip.so for gfx950 (cross-compiles without a GPU).  Usage: bfsr_amd/csrc/build.sh [--clean]
set -e
cd "$(dirname "$0")"
OUT=../lib
if [ "$1" = "--clean" ]; then rm -rf build "$OUT"/libbfsr_hip.so; fi
mkdir -p "$OUT" build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
pids=()
for f in conv_mfma conv_f16 conv_bf16x3 conv_x3s conv_h2s conv_chain conv_up2_h2t conv_up4_h2t conv1x1 flow_ops coupling coupling_tail resample linf_ops linf_mlp metrics range_check; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ ../../include/bfsr_hip.h -nt build/$f.o ] || [ launch_util.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/conv_mfma.o build/conv_f16.o build/conv_bf16x3.o build/conv_x3s.o build/conv_h2s.o build/conv_chain.o build/conv_up2_h2t.o build/conv_up4_h2t.o build/conv1x1.o build/flow_ops.o build/coupling.o build/coupling_tail.o build/resample.o build/linf_ops.o build/linf_mlp.o build/metrics.o build/range_check.o -o "$OUT/libbfsr_hip.so"
echo "built $OUT/libbfsr_hip.so"
