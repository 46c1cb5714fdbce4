OUT=gpurun_out/r06o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "h2s" > $OUT/pytest_h2s.txt 2>&1; tail -5 $OUT/pytest_h2s.txt
timeout 900 python -m pytest tests/test_linf_gpu.py tests/test_determinism_gpu.py -x -q > $OUT/pytest_linf.txt 2>&1; tail -4 $OUT/pytest_linf.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line"
$B --config 5 > $OUT/cfg5_res1.json 2> $OUT/cfg5_res1.err
BFSR_H2S_RES=0 $B --config 5 > $OUT/cfg5_res0.json 2> $OUT/cfg5_res0.err
python tools/exp/linf_keys.py --config 5 --top 12 > $OUT/keys_cfg5.txt 2> $OUT/keys_cfg5.err
for f in cfg5_res1 cfg5_res0; do python - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"])
PY
done
head -14 $OUT/keys_cfg5.txt
