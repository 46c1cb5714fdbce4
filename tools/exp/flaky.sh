# failure counts of the batch-sharding test under different switches (one box): bash tools/exp/flaky.sh [runs] [VAR=val ...]
N=${1:-12}; shift
run() { n=0; for i in $(seq $N); do if ! env "$@" timeout 300 python -m pytest tests/test_srflow_gpu.py -x -q -k roundtrip_and_batch 2>&1 | grep -q "1 passed"; then n=$((n+1)); fi; done; echo "$* : $n / $N failed"; }
if [ $# -gt 0 ]; then run "$@"; else run BFSR_OVERLAP=1; run BFSR_OVERLAP=0; fi
