#!/bin/bash
# PMC wave-cycle breakdown of conv_chain_kernel on the dense-block convs (chain-of-one); run through gpurun.  usage: tools/exp/pmc_chain.sh <tag> B H
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc_chain}; B=${2:-16}; H=${3:-256}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/exp/chain_one.py $B $H 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES \
   --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM \
   --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
python $R/tools/exp/pmc_sum.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
