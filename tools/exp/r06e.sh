OUT=gpurun_out/r06e; mkdir -p $OUT
python tools/exp/pass_jitter.py --batch 16 --passes 40 > $OUT/jitter_b16.txt 2>&1
cat $OUT/jitter_b16.txt | grep -v "^UNet"
