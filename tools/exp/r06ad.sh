#!/bin/bash
# round 6: h2 pooling / up-sampling kernels of the learned priors (BFSR_PRIOR_GLUE=fused|launches): tests + alternating bench processes.  gpurun: bash tools/exp/r06ad.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06ad; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
(timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_linf_gpu.py tests/test_srflow_gpu.py -m gpu -x -q 2>&1 | tail -4) > $OUT/${TAG}_tests.txt; cat $OUT/${TAG}_tests.txt
for rep in 1 2; do
  for g in fused launches; do
    for cfg in 5 4 3; do
      st=10; [ $cfg = 4 ] && st=4
      BFSR_PRIOR_GLUE=$g python bench.py --config $cfg --steps $st --warmup 3 --no-cpu-baseline --no-fp32-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
fam={e['kernel']:e['ms_per_step'] for e in (d.get('roofline_by_symbol') or [])}
print('rep $rep glue %-8s cfg $cfg: %8.3f ms  %.2f MPix/s' % ('$g', d['ms_per_step'], d['value']), {k:v for k,v in fam.items() if k in ('resize','h2_pack','h2_unpack','maxpool2','resize_h2','maxpool2_h2')})"
    done
  done
done > $OUT/${TAG}_glue.txt 2>&1
cat $OUT/${TAG}_glue.txt
(python tools/env_ab.py BFSR_PRIOR_GLUE launches fused 2>&1 | tail -3) > $OUT/${TAG}_ab_cfg2.txt; cat $OUT/${TAG}_ab_cfg2.txt
