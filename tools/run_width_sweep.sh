# Runs ON THE GPU BOX: one-GPU bench.py at slab widths x steps per pass {1 (k_step), 2, 3, 4}, one line each into gpurun_out/width_sweep.log
out=gpurun_out/width_sweep.log; : > $out
for w in ${WIDTHS:-288 416 544 800 1056 2080 4096}; do
  for depth in 0 2 3 4; do
    fuse="--fuse 2 --fuse-depth $depth"; [ $depth = 0 ] && fuse="--fuse 0"
    echo -n "$w x 4096 steps/pass $depth: " >> $out
    timeout -k 10 200 python3 bench.py --nx $w --ny 4096 $fuse --cpu-steps 0 --steps 408 --warmup 24 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('units %5d columns/unit %3d  %7.2f us/step %6.1f GLUPS'%(c['fuse_units'], c['fuse_chunk'], d['ms_per_step']*1e3, d['value']/1e3))
    else: print(l[:200])
" >> $out
  done
done
cat $out
