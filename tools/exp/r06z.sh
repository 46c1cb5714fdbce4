#!/bin/bash
# round 6: XCD-aware block order in the register-staged convs (conv_f16, conv_bf16x3 x3, conv_mfma x2, conv1x1) against the plain block order
# (tools/exp/libchain_plainorder.so = HEAD~ of those four files), alternating processes on one box.  gpurun: bash tools/exp/r06z.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06z; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
(timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_linf_gpu.py tests/test_srflow_gpu.py -m gpu -x -q 2>&1 | tail -3) > $OUT/${TAG}_tests.txt; cat $OUT/${TAG}_tests.txt
for rep in 1 2; do
  for lib in default plainorder; do
    for cfg in 2 3 4 5; do
      L=$R/bfsr_amd/lib/libbfsr_hip.so; [ $lib != default ] && L=$R/tools/exp/libchain_$lib.so
      st=10; [ $cfg = 4 ] && st=4
      BFSR_HIP_LIB=$L python bench.py --config $cfg --steps $st --warmup 3 --no-cpu-baseline --no-fp32-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
fam={e['kernel']:e['ms_per_step'] for e in (d.get('roofline_by_symbol') or [])}
print('rep $rep lib %-10s cfg $cfg: %8.3f ms' % ('$lib', d['ms_per_step']), {k:v for k,v in fam.items() if 'bf16x3' in k or 'conv_f16' in k or 'conv_mfma' in k or 'conv1x1' in k})"
    done
  done
done > $OUT/${TAG}_block_order.txt 2>&1
cat $OUT/${TAG}_block_order.txt
