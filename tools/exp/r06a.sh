OUT=gpurun_out/r06a; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line"
$B --config 4 --batch 8  > $OUT/cfg4_b8.json 2> $OUT/cfg4_b8.err
$B --config 4            > $OUT/cfg4_b64.json 2> $OUT/cfg4_b64.err
$B --config 5 --batch 16 > $OUT/cfg5_b16.json 2> $OUT/cfg5_b16.err
$B --config 5            > $OUT/cfg5_b128.json 2> $OUT/cfg5_b128.err
$B --config 2            > $OUT/cfg2.json 2> $OUT/cfg2.err
BFSR_OVERLAP=0 python tools/profile_keys.py --scale 8 --batch 8 --lr 96 --top 60 2>/dev/null | grep -v "^UNet" > $OUT/keys_cfg4_b8_no_overlap.txt
BFSR_OVERLAP=0 python tools/profile_keys.py --scale 8 --batch 64 --lr 96 --top 60 2>/dev/null | grep -v "^UNet" > $OUT/keys_cfg4_b64_no_overlap.txt
BFSR_OVERLAP=0 python tools/profile_keys.py --top 60 2>/dev/null | grep -v "^UNet" > $OUT/keys_cfg2_no_overlap.txt
python tools/exp/linf_keys.py --help > $OUT/linf_keys_help.txt 2>&1
tail -c 600 $OUT/cfg4_b8.json; echo; tail -c 300 $OUT/cfg4_b8.err
