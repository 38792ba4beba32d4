for rep in 1 2; do for cfg in "1.0 2.2" "1.25 1.6" "1.4 1.4" "1.25 2.2"; do set -- $cfg; for a in "--nx 544" "--nx 1056" "--nx 2080" "--nx 4096" "--nx 4096 --ny 2048 --dtype float64"; do
echo -n "# beta $1 alpha $2 $a: "; WT_BETA=$1 WT_ALPHA=$2 python3 bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 $a 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   %.2f us/step %.1f GLUPS units %d depth %d'%(d['ms_per_step']*1e3, d['value']/1e3, d['config']['fuse_units'], d['config']['fuse_depth']))
    else: print(l[:200])
"; done; done; done
