#!/bin/bash
# PMC passes over the conv_h2s micro-benchmark (pmc_sum.py prints the LAST dispatch per kernel symbol = the widest conv of the list).
# Every rocprofv3 run is wrapped in `timeout`: a counter set the tool rejects aborts and then hangs in finalisation.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_h2s; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/exp/h2s_bench.py $@"
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
python $R/tools/exp/pmc_sum.py $OUT/p1 $OUT/p2 $OUT/p3 | grep -A32 "conv3x3_h2s" | head -70
find $OUT -name "*.csv" -size +2M -delete
