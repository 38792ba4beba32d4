# Runs ON THE GPU BOX: the measured refinement of the marching units (option "tune") against the modelled cut, on the whole lattice, the slabs of
# the real 2/4/8-way splits (equal widths, then cut by measured cost), the width proxies, fp64 and fast_math
for t in 0 1; do
  echo "== WT_TUNE=$t"
  WT_TUNE=$t WT_BALANCE_ROUNDS=$((t * 4)) python3 tools/r3_slab_costs.py 2>&1 | grep -v amdgpu.ids
  for a in "--nx 544" "--nx 1056" "--dtype float64" "--dtype float64 --nx 4096 --ny 2048" "--fast-math 1"; do echo -n "   $a: "; WT_TUNE=$t python3 bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 --pmc-traffic 0 --fast-math 0 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f us/step %.1f GLUPS'%(d['ms_per_step']*1e3, d['value']/1e3))"; done
done
