OUT=gpurun_out/r06c; mkdir -p $OUT
python tools/exp/host_profile_linf.py --batch 16 > $OUT/host_cfg5_b16.txt 2> $OUT/host_cfg5_b16.err
python -m pytest tests/test_range_guard_gpu.py -x -q > $OUT/pytest_guard.txt 2>&1
tail -3 $OUT/pytest_guard.txt; head -120 $OUT/host_cfg5_b16.txt
