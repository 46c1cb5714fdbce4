#!/bin/bash
# round 6: compact x4 hand-over as class rows + up4 fetched at the top of the epilogue by its own instantiation: tests, config-4 bench + kernel stats + PMC traffic, per-workgroup chain trace.  gpurun: bash tools/exp/r06u.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06v; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_srflow_gpu.py tests/test_host_logic.py -m gpu -q -k "up4 or compact or config4 or 8x or abi" 2>&1 | tail -4) > $OUT/${TAG}_tests.txt
cat $OUT/${TAG}_tests.txt
(python tools/env_ab.py BFSR_UP4C 0 1 --scale 8 --batch 64 --lr 96 2>&1 | tail -4) > $OUT/${TAG}_ab_up4c.txt; cat $OUT/${TAG}_ab_up4c.txt
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_cfg4_bench.json 2> $OUT/cfg4.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- $B --config 4 > $OUT/${TAG}_cfg4_bench_under_rocprof.json 2> $OUT/stats_cfg4.err
f=$(find $OUT/stats_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_cfg4_kernel_stats.csv
rm -rf $OUT/stats_cfg4
bash $R/tools/pmc_traffic_cfg.sh 4 $TAG
rm -rf $OUT/pmc_fetch_4 $OUT/pmc_write_4
cd $R
python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line > $OUT/${TAG}_cfg2_bench.json 2> /dev/null
python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_cfg3_bench.json 2> /dev/null
python - <<P
import json
for c in (2, 3):
    d = json.loads(open("$OUT/${TAG}_cfg%d_bench.json" % c).read().strip().splitlines()[-1]); print("cfg", c, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
P
head -5 $OUT/${TAG}_cfg4_kernel_stats.csv | cut -c1-160
python - <<P
import json
d = json.loads(open("$OUT/${TAG}_cfg4_bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
t = json.load(open("$OUT/${TAG}_pmc_traffic_cfg4.json"))["kernels"]
for k, v in t.items():
    if "up4" in k or "1024, 64, 384" in k: print(k, round(v["fetch_bytes_raw"] * 2 / 1e9, 1), round(v["write_bytes"] / 1e9, 1), round(v["hbm_bytes_per_launch"] / 1e9, 1))
P
