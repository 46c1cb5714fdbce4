#!/bin/bash
# fused LINF conditioning kernel: unit tests + A/B of bench configs 3 and 5 against the unfused path
python -m pytest tests/test_linf_gpu.py -q -x -k "linf_mlp_fused or golden_e2e or fp16_mfma" 2>&1 | grep -v "^UNet\|amdgpu" | tail -5
for m in fused unfused; do
  for c in 3 5; do
    BFSR_LINF_MLP=$m python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m cfg$c', d['value'], d['ms_per_step'])"
  done
done
python tools/linf_bench.py --scales 4 --steps 2 > /dev/null 2>&1
