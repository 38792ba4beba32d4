for prio in 0 1 2 0 1 2; do for a in "--nx 544 --fuse-depth 3" "--nx 544 --fuse-depth 4" "--nx 1056 --fuse-depth 4" "--nx 4096"; do
echo "# prio $prio $a"; WT_PRIO=$prio python3 bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 $a 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   %.2f us/step %.1f GLUPS units %d'%(d['ms_per_step']*1e3, d['value']/1e3, d['config']['fuse_units']))
    else: print(l[:200])
"; done; done
WT_PRIO=1 python3 tools/unit_clocks.py 544 4096 3 1 2>&1 | grep -v amdgpu.ids | head -12
