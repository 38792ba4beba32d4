for w in 288 544 1056 2080 4096; do for depth in 0 2 3 4; do
fuse="--fuse 2 --fuse-depth $depth"; [ $depth = 0 ] && fuse=""
echo -n "fp64 $w x 4096 forced steps/pass $depth (0 = automatic): "; python3 bench.py --dtype float64 --nx $w --ny 4096 $fuse --cpu-steps 0 --steps 204 --warmup 24 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('depth %d units %5d columns/unit %3d  %7.2f us/step %6.1f GLUPS'%(c['fuse_depth'], c['fuse_units'], c['fuse_chunk'], d['ms_per_step']*1e3, d['value']/1e3))
    else: print(l[:200])
"; done; done
