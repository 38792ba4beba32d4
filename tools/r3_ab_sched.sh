# Runs ON THE GPU BOX: the scheduler variant (-mllvm -amdgpu-sched-strategy=max-ilp) against the default build, interleaved
for rep in 1 2 3; do for lib in lib_base.so lib_maxilp.so; do
  for a in "" "--nx 544" "--dtype float64 --nx 4096 --ny 2048"; do echo -n "$lib $a: "; WT_AB_LIB=$lib python3 tools/ab_bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 --pmc-traffic 0 --fast-math 0 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f us/step %.1f GLUPS'%(d['ms_per_step']*1e3, d['value']/1e3))"; done; done; done
