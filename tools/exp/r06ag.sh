#!/bin/bash
# round 6, final build (after the glue kernels): full GPU suite, config-2 evidence (tools/profile_light.sh), default bench lines of configs 3 / 4 / 5, the per-rank shards of the
# strong-scaling configurations.  gpurun: bash tools/exp/r06x.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06ag; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
(python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/${TAG}_gputest.txt
(python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2) > $OUT/${TAG}_smoke.txt
bash tools/profile_light.sh $TAG > /dev/null 2>&1
cd $R
for c in 3 4 5; do python bench.py --config $c > $OUT/${TAG}_cfg${c}_bench.json 2> /dev/null; done
python bench.py --config 4 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_cfg4_b8_bench.json 2> /dev/null
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_cfg4_b64_bench.json 2> /dev/null
python bench.py --config 5 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_cfg5_b16_bench.json 2> /dev/null
python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_cfg5_b128_bench.json 2> /dev/null
BFSR_OVERLAP=0 python tools/profile_keys.py --scale 8 --batch 64 --lr 96 --top 60 2>/dev/null | grep -v "^UNet" > $OUT/${TAG}_keys_cfg4_b64_no_overlap.txt
python tools/exp/linf_keys.py --config 5 --top 45 > $OUT/${TAG}_keys_cfg5.txt 2>/dev/null
python tools/exp/linf_keys.py --config 3 --top 30 > $OUT/${TAG}_keys_cfg3.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for c in 4 5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line > $OUT/${TAG}_cfg${c}_bench_under_rocprof.json 2> $OUT/stats_cfg$c.err
  f=$(find $OUT/stats_cfg$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_cfg${c}_kernel_stats.csv
  rm -rf $OUT/stats_cfg$c
done
cd $R
cat $OUT/${TAG}_gputest.txt $OUT/${TAG}_smoke.txt
python - <<P
import json, glob
for f in sorted(glob.glob("$OUT/${TAG}_cfg*_bench.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r["kernel"][:50], r["frac"], (d.get("parity") or {}).get("max_abs_sr"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
P
