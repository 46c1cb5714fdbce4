#!/bin/bash
# PMC wave-cycle breakdown of the x3 conv kernels on the RDB shapes (diagnostic; run through gpurun).
# usage: tools/exp/pmc_conv.sh <tag> [conv_bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---x3 --x3tunes=0 --tunes=0 --only=rdb.conv1 --only=rdb.conv4 --only=rdb.conv5 --only=hoist}"
rocprofv3 -L > $OUT/counters.txt 2>&1
python $R/tools/conv_bench.py $ARGS > $OUT/plain.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES \
   --kernel-trace --output-format csv -d $OUT/p1 -- python $R/tools/conv_bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $OUT/p2 -- python $R/tools/conv_bench.py $ARGS > $OUT/p2.log 2>&1
python $R/tools/exp/pmc_sum.py $OUT/p1 $OUT/p2 > $OUT/summary.txt 2>&1
cat $OUT/plain.log; cat $OUT/summary.txt
# keep only the small files
find $OUT -name "*.csv" -size +2M -delete
