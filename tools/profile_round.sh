#!/bin/bash
# End-of-round evidence (run through gpurun): kernel-trace stats of the bench configs, PMC traffic passes and matrix-pipe counters of
# the headline config.  Every rocprofv3 run sits under `timeout` (a rejected counter set aborts and then hangs in finalisation).
# Usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>/...
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line"
for c in 2 3 4 5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- $B --config $c > $OUT/stats_cfg$c.json 2> $OUT/stats_cfg$c.err
  f=$(find $OUT/stats_cfg$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_cfg${c}_kernel_stats.csv
done
B1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line"
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_fetch.json timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
BFSR_OVERLAP=0 BFSR_KEYLOG=$OUT/keys_write.json timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B1 > /dev/null 2> $OUT/pmc_write.err
python $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/keys_fetch.json $OUT/pmc_write $OUT/keys_write.json > $OUT/${TAG}_pmc_traffic.json 2> $OUT/pmc_traffic.err
BFSR_KEYLOG=$OUT/keys_fetch3.json timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch3 -- $B1 --config 3 > /dev/null 2> $OUT/pmc_fetch3.err
BFSR_KEYLOG=$OUT/keys_write3.json timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write3 -- $B1 --config 3 > /dev/null 2> $OUT/pmc_write3.err
python $R/tools/pmc_traffic.py $OUT/pmc_fetch3 $OUT/keys_fetch3.json $OUT/pmc_write3 $OUT/keys_write3.json > $OUT/${TAG}_pmc_traffic_cfg3.json 2> $OUT/pmc_traffic3.err
BFSR_OVERLAP=0 timeout 900 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
   --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B1 > /dev/null 2> $OUT/pmc_sq.err
python $R/tools/exp/pmc_sum.py $OUT/pmc_sq > $OUT/${TAG}_pmc_mfma_busy.txt 2>&1
BFSR_OVERLAP=0 python $R/tools/profile_keys.py --top 45 2>/dev/null | grep -v "^UNet" > $OUT/${TAG}_keys_cfg2_no_overlap.txt
# kernel ablations / micro-benchmarks quoted in DESIGN.md (tools/exp/libabl.so = the library built with -DBFSR_X3S_ABL -DBFSR_H2S_ABL)
if [ -f $R/tools/exp/libabl.so ]; then
  (cd $R && timeout 300 python tools/exp/x3s_abl.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_b_x3s_ablation.txt)
  (cd $R && TUNES="0 -3 -11 -4" timeout 600 bash tools/exp/h2s_abl.sh > $OUT/${TAG}_d_h2s_ablation.txt 2>&1)
fi
(cd $R && timeout 300 python tools/exp/h2s_bench.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_d_h2s_bench.txt)
(cd $R && timeout 300 python tools/exp/h2x_bench.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_e_h2x_vs_x3s.txt)
(cd $R && BFSR_SPLIT=bf16x3 timeout 600 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-line > $OUT/${TAG}_cfg2_bench_bf16x3.json 2>/dev/null)
find $OUT -name "*.csv" -size +3M -delete
ls -la $OUT | head -40; tail -3 $OUT/pmc_traffic.err
