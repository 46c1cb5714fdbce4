R=$PWD; OUT=$R/gpurun_out/r06n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace2 -- python $R/bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-line > $OUT/cfg2_rocprof.json 2> $OUT/cfg2_rocprof.err
cd $R
python tools/exp/gap_report.py $OUT/trace2 > $OUT/gaps_cfg2.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/gaps_cfg2.txt
