export BFSR_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/libabl.so
for t in ${TUNES:-0 -3 -11 -8}; do echo "== tune $t"; timeout 120 python tools/exp/h2s_bench.py 128 128 128 $t 2>&1 | grep -v amdgpu; done
