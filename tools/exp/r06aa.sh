#!/bin/bash
# round 6: nt stores for linf_mlp's affine_info rows (tools/exp/libchain_mlpnt.so: linf_mlp.hip with -DBFSR_MLP_NT=1) against the default, alternating processes.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06aa; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for rep in 1 2 3; do
  for lib in default mlpnt; do
    for cfg in 5 3; do
      L=$R/bfsr_amd/lib/libbfsr_hip.so; [ $lib != default ] && L=$R/tools/exp/libchain_$lib.so
      BFSR_HIP_LIB=$L python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
fam={e['kernel']:e['ms_per_step'] for e in (d.get('roofline_by_symbol') or [])}
print('rep $rep lib %-8s cfg $cfg: %8.3f ms' % ('$lib', d['ms_per_step']), {k:v for k,v in fam.items() if 'linf' in k})"
    done
  done
done > $OUT/${TAG}_mlp_nt.txt 2>&1
cat $OUT/${TAG}_mlp_nt.txt
