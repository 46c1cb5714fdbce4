#!/bin/bash
# round 6: PMC traffic of config 2 on the build of record (tools/pmc_traffic.py knows the new kernel names now) + the driver's own bench command.  gpurun: bash tools/exp/r06aj.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06aj; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_driver_style_bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
B1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line"
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_fetch.json timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_write.json timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B1 > /dev/null 2> $OUT/pmc_write.err
python $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/keys_fetch.json $OUT/pmc_write $OUT/keys_write.json > $OUT/${TAG}_pmc_traffic.json 2> $OUT/pmc_traffic.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write; cat $OUT/pmc_traffic.err; wc -c $OUT/${TAG}_pmc_traffic.json
python -c "
import json; d=json.loads(open('$OUT/${TAG}_driver_style_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), d['parity']['max_abs_sr'], d['cpu_baseline']['value'], d['fallbacks'])"
