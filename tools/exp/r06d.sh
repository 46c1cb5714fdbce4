R=$PWD; OUT=$R/gpurun_out/r06d; mkdir -p $OUT
python tools/exp/linf_loop.py --batch 16 --passes 10 > $OUT/loop_b16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/tools/exp/linf_loop.py --batch 16 --passes 6 > $OUT/loop_b16_rocprof.txt 2>&1
cd $R
python tools/exp/gap_report.py $OUT/trace > $OUT/gaps_cfg5_b16.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/loop_b16.txt | tail -2; tail -1 $OUT/loop_b16_rocprof.txt; cat $OUT/gaps_cfg5_b16.txt
