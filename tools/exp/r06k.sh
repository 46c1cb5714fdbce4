bash tools/profile_light.sh r06k > /dev/null 2>&1
OUT=gpurun_out/r06k
for c in 3 4 5; do python bench.py --config $c > $OUT/r06k_cfg${c}_bench.json 2> $OUT/cfg$c.err; done
for c in 2 3 4 5; do python - <<PY
import json
d=json.loads(open("$OUT/r06k_cfg${c}_bench.json").read().strip().splitlines()[-1]); print($c, d["value"], d["ms_per_step"], d.get("parity"), d["roofline"].get("frac"), d.get("cpu_baseline",{}).get("value"))
PY
done
ls $OUT
