#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (one counter per pass) over one bench configuration, dispatches mapped to launch shapes through
# BFSR_KEYLOG.  Usage: tools/pmc_traffic_cfg.sh <config> <tag>  -> gpurun_out/<tag>/<tag>_pmc_traffic_cfg<config>.json
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$1; TAG=${2:-r02}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B1="python $R/bench.py --config $C --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line"
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_fetch_$C.json timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$C -- $B1 > /dev/null 2> $OUT/pmc_fetch_$C.err
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_write_$C.json timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$C -- $B1 > /dev/null 2> $OUT/pmc_write_$C.err
python $R/tools/pmc_traffic.py $OUT/pmc_fetch_$C $OUT/keys_fetch_$C.json $OUT/pmc_write_$C $OUT/keys_write_$C.json > $OUT/${TAG}_pmc_traffic_cfg$C.json 2> $OUT/pmc_traffic_$C.err
tail -2 $OUT/pmc_traffic_$C.err; find $OUT -name "*.csv" -size +3M -delete
