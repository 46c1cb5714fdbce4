#!/bin/bash
# Mid-round evidence for the headline config only (run through gpurun): kernel-trace stats, PMC traffic (separate FETCH / WRITE passes),
# per-launch-shape table without the side stream, plain bench line.  Usage: tools/profile_light.sh <tag> -> gpurun_out/<tag>/...
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04a}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2 -- $B --config 2 > $OUT/${TAG}_cfg2_bench_under_rocprof.json 2> $OUT/stats_cfg2.err
f=$(find $OUT/stats_cfg2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_cfg2_kernel_stats.csv
B1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line"
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_fetch.json timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_write.json timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B1 > /dev/null 2> $OUT/pmc_write.err
python $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/keys_fetch.json $OUT/pmc_write $OUT/keys_write.json > $OUT/${TAG}_pmc_traffic.json 2> $OUT/pmc_traffic.err
BFSR_OVERLAP=0 python $R/tools/profile_keys.py --top 60 2>/dev/null | grep -v "^UNet" > $OUT/${TAG}_keys_cfg2_no_overlap.txt
(cd $R && python bench.py --config 2 --steps 10 --warmup 3 > $OUT/${TAG}_cfg2_bench.json 2>/dev/null)
find $OUT -name "*.csv" -size +3M -delete
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats_cfg2
ls -la $OUT | head; tail -3 $OUT/pmc_traffic.err; cat $OUT/${TAG}_keys_cfg2_no_overlap.txt | head -50
