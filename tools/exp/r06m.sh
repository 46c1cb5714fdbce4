OUT=gpurun_out/r06m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_linf_gpu.py -x -q > $OUT/pytest_linf.txt 2>&1; tail -4 $OUT/pytest_linf.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line"
for t in 128 64; do
BFSR_MLP_TILE=$t $B --config 5 > $OUT/cfg5_t$t.json 2> $OUT/cfg5_t$t.err
BFSR_MLP_TILE=$t $B --config 3 > $OUT/cfg3_t$t.json 2> $OUT/cfg3_t$t.err
done
python tools/exp/linf_keys.py --config 5 --top 10 > $OUT/keys_cfg5.txt 2> $OUT/keys_cfg5.err
python tools/exp/linf_keys.py --config 3 --top 10 > $OUT/keys_cfg3.txt 2> $OUT/keys_cfg3.err
for f in cfg5_t128 cfg5_t64 cfg3_t128 cfg3_t64; do python - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"])
PY
done
grep mlp $OUT/keys_cfg5.txt $OUT/keys_cfg3.txt
