for rep in 1 2; do for a in "--nx 2080" "--nx 3000 --ny 3000" "--nx 4096" "--nx 4096 --ny 2048 --dtype float64"; do for r in 1 2; do
echo -n "# rounds $r $a: "; WT_MARCH_ROUNDS=$r python3 bench.py --ny 4096 --cpu-steps 0 --fast-math 0 --steps 408 --warmup 24 $a 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   %.2f us/step %.1f GLUPS units %d cols/unit %d'%(d['ms_per_step']*1e3, d['value']/1e3, d['config']['fuse_units'], d['config']['fuse_chunk']))
    else: print(l[:200])
"; done; done; done
