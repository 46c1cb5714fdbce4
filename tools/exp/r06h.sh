OUT=gpurun_out/r06h; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line"
$B --config 2            > $OUT/cfg2.json 2> $OUT/cfg2.err
BFSR_WIDE=0 $B --config 2 > $OUT/cfg2_wide0.json 2> $OUT/cfg2_wide0.err
$B --config 4 --batch 8  > $OUT/cfg4_b8.json 2> $OUT/cfg4_b8.err
$B --config 4            > $OUT/cfg4_b64.json 2> $OUT/cfg4_b64.err
BFSR_OVERLAP=0 python tools/profile_keys.py --top 60 2>/dev/null | grep -v "^UNet" > $OUT/keys_cfg2_no_overlap.txt
BFSR_OVERLAP=0 python tools/profile_keys.py --scale 8 --batch 8 --lr 96 --top 60 2>/dev/null | grep -v "^UNet" > $OUT/keys_cfg4_b8_no_overlap.txt
for f in cfg2 cfg2_wide0 cfg4_b8 cfg4_b64; do python - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d.get("parity"))
PY
done
grep -E "wide|flow|h2_pack" $OUT/keys_cfg2_no_overlap.txt $OUT/keys_cfg4_b8_no_overlap.txt
