#!/bin/bash
# round 6, after the compact x4 hand-over became the default: new tests, the N>1 launch form at world 1, config-4 evidence (bench line, kernel stats,
# PMC traffic), matrix-pipe busy fraction of the shipped kernels at config 2.  Run through gpurun: tools/exp/r06q.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r06q; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
(python -m pytest tests/test_srflow_gpu.py tests/test_dist_rccl.py -m gpu -q -k "compact or config4 or rccl or gather" 2>&1 | tail -5) > $OUT/${TAG}_newtests.txt
BFSR_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-line > $OUT/${TAG}_torchrun_world1_bench.json 2> $OUT/torchrun.err
python bench.py --config 4 > $OUT/${TAG}_cfg4_bench.json 2> $OUT/cfg4.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- $B --config 4 > $OUT/${TAG}_cfg4_bench_under_rocprof.json 2> $OUT/stats_cfg4.err
f=$(find $OUT/stats_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_cfg4_kernel_stats.csv
rm -rf $OUT/stats_cfg4
bash $R/tools/pmc_traffic_cfg.sh 4 $TAG
rm -rf $OUT/pmc_fetch_4 $OUT/pmc_write_4
BFSR_OVERLAP=0 timeout 900 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
   --kernel-trace --output-format csv -d $OUT/pmc_busy -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line > /dev/null 2> $OUT/pmc_busy.err
python $R/tools/exp/pmc_sum.py $OUT/pmc_busy > $OUT/${TAG}_pmc_mfma_busy.txt 2>&1
rm -rf $OUT/pmc_busy
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/${TAG}_newtests.txt; tail -c 300 $OUT/${TAG}_torchrun_world1_bench.json; tail -3 $OUT/torchrun.err
python - <<P
import json
for f in ("$OUT/${TAG}_cfg4_bench.json",):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("parity"))
P
grep -A 12 "conv_chain" $OUT/${TAG}_pmc_mfma_busy.txt | head -30
