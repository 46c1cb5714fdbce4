cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for n in 2 8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp$n -- python $R/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-fp32-line > /dev/null 2>&1
  f=$(find /tmp/cp$n -name "*kernel_stats.csv" | head -1); echo "steps=$n: $(grep -i copyBuffer $f | cut -d, -f1-3)"
done
