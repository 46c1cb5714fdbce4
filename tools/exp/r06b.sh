OUT=gpurun_out/r06b; mkdir -p $OUT
python tools/exp/linf_keys.py --config 5 --batch 16 --top 50 > $OUT/keys_cfg5_b16.txt 2> $OUT/keys_cfg5_b16.err
python -m pytest tests/test_conv_chain.py tests/test_range_guard_gpu.py -x -q > $OUT/pytest_chain.txt 2>&1
tail -3 $OUT/pytest_chain.txt; head -40 $OUT/keys_cfg5_b16.txt
