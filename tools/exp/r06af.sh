cd $GRAFT_REPO_ROOT
for rep in 1 2; do for t in 32 8; do for cfg in "5" "5 --batch 16" "3"; do
BFSR_PRIOR_MIN_TILES=$t python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep tiles %-3s cfg %-14s: %8.3f ms' % ('$t', '$cfg', d['ms_per_step']))"
done; done; done
