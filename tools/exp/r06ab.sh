#!/bin/bash
# round 6: fused LINF harness glue (BFSR_LINF_GLUE=fused|launches): tests + alternating bench processes.  gpurun: bash tools/exp/r06ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06ab; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
(timeout 1200 python -m pytest tests/test_linf_gpu.py tests/test_hip_ops.py -m gpu -x -q 2>&1 | tail -4) > $OUT/${TAG}_tests.txt; cat $OUT/${TAG}_tests.txt
for rep in 1 2; do
  for g in fused launches; do
    for cfg in 5 3; do
      BFSR_LINF_GLUE=$g python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
fam={e['kernel']:e['ms_per_step'] for e in (d.get('roofline_by_symbol') or [])}
print('rep $rep glue %-8s cfg $cfg: %8.3f ms  %.2f MPix/s' % ('$g', d['ms_per_step'], d['value']), {k:v for k,v in fam.items() if k in ('resize','axpb_clamp','patch_fold','patch_unfold','linf_fold_skip','linf_prep_residual','linf_prep_down')})"
    done
  done
done > $OUT/${TAG}_glue.txt 2>&1
cat $OUT/${TAG}_glue.txt
python bench.py --config 5 --steps 3 --warmup 1 > $OUT/${TAG}_cfg5_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/${TAG}_cfg5_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity'))"
