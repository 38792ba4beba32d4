# Runs ON THE GPU BOX: planner constants (WT_BETA, WT_ALPHA) against the slowest slab of the real 8- and 4-way splits, the 544 / 1056 proxies and 4096^2
for cfg in "1.25 1.6" "1.1 1.4" "1.0 1.2" "1.25 1.2" "1.1 1.0" "1.0 0.8"; do set -- $cfg
  echo "== beta $1 alpha $2"
  WT_BETA=$1 WT_ALPHA=$2 python3 tools/r3_slab_costs.py 2>&1 | grep -E "whole|slowest|per-slab"
  for a in "--nx 544" "--nx 1056"; do echo -n "   proxy $a: "; WT_BETA=$1 WT_ALPHA=$2 python3 bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 --pmc-traffic 0 --fast-math 0 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f us/step'%(d['ms_per_step']*1e3))"; done
done
