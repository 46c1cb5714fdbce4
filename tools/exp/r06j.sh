OUT=gpurun_out/r06j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "h2s" > $OUT/pytest_h2s.txt 2>&1; tail -5 $OUT/pytest_h2s.txt
timeout 900 python -m pytest tests/test_linf_gpu.py -x -q > $OUT/pytest_linf.txt 2>&1; tail -5 $OUT/pytest_linf.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line"
$B --config 5 > $OUT/cfg5_mt2.json 2> $OUT/cfg5_mt2.err
BFSR_H2S_MT=1 $B --config 5 > $OUT/cfg5_mt1.json 2> $OUT/cfg5_mt1.err
$B --config 5 --batch 16 > $OUT/cfg5_b16_mt2.json 2> $OUT/cfg5_b16.err
python tools/exp/linf_keys.py --config 5 --top 14 > $OUT/keys_cfg5.txt 2> $OUT/keys_cfg5.err
for f in cfg5_mt2 cfg5_mt1 cfg5_b16_mt2; do python - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"])
PY
done
head -16 $OUT/keys_cfg5.txt
