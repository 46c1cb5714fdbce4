OUT=gpurun_out/r06l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_linf_gpu.py tests/test_hip_ops.py -x -q -k "linf or flow" > $OUT/pytest_linf.txt 2>&1; tail -4 $OUT/pytest_linf.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line"
$B --config 5 > $OUT/cfg5.json 2> $OUT/cfg5.err
$B --config 3 > $OUT/cfg3.json 2> $OUT/cfg3.err
python tools/exp/linf_keys.py --config 5 --top 12 > $OUT/keys_cfg5.txt 2> $OUT/keys_cfg5.err
for f in cfg5 cfg3; do python - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"])
PY
done
head -14 $OUT/keys_cfg5.txt
python tools/env_ab.py BFSR_LANES 1 0 --rounds 5 --block 6 > $OUT/ab_lanes_cfg2.txt 2>&1; tail -4 $OUT/ab_lanes_cfg2.txt
python tools/env_ab.py BFSR_LANES 1 0 --scale 8 --batch 8 --lr 96 --rounds 5 --block 6 > $OUT/ab_lanes_cfg4b8.txt 2>&1; tail -4 $OUT/ab_lanes_cfg4b8.txt
