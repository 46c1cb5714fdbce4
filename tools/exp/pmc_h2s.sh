#!/bin/bash
# PMC passes over the conv_h2s micro-benchmark (last dispatch = the 192->64 conv5; per-shape via Grid_Size is the same, so look at the LAST dispatch)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_h2s; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/exp/h2s_bench.py $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
python $R/tools/exp/pmc_sum.py $OUT/p1 $OUT/p2 $OUT/p3 | grep -A32 "conv3x3_h2s" | head -40
tail -3 $OUT/p2.log
find $OUT -name "*.csv" -size +2M -delete
