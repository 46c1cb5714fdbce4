#!/bin/bash
# round 6: cache policy of the large streamed conv outputs (default / nt / sc1), alternating processes on one box.  gpurun: bash tools/exp/r06w.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06w; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for rep in 1 2; do
  for lib in default out2 out16; do
    for cfg in 2 4; do
      L=$R/bfsr_amd/lib/libbfsr_hip.so; [ $lib != default ] && L=$R/tools/exp/libchain_$lib.so
      st=10; [ $cfg = 4 ] && st=4
      BFSR_HIP_LIB=$L python bench.py --config $cfg --steps $st --warmup 3 --no-cpu-baseline --no-fp32-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rep $rep lib %-8s cfg $cfg: %8.3f ms  %s %.4f %.3f ms' % ('$lib', d['ms_per_step'], r['kernel'][:40], r['frac'], r['avg_launch_ms']), [ (x['kernel'][:34], x['avg_launch_ms']) for x in d['roofline_next_kernels'][:3]])"
    done
  done
done > $OUT/${TAG}_out_policy.txt 2>&1
cat $OUT/${TAG}_out_policy.txt
