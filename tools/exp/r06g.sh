OUT=gpurun_out/r06g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_coupling_wide.py -x -q > $OUT/pytest_wide.txt 2>&1
tail -25 $OUT/pytest_wide.txt
timeout 900 python -m pytest tests/test_srflow_gpu.py tests/test_ref_goldens_gpu.py -x -q > $OUT/pytest_srflow.txt 2>&1
tail -15 $OUT/pytest_srflow.txt
