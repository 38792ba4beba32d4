#!/bin/bash
# Runs ON THE GPU BOX: A/B of the chain blocks (step_chain.hpp) against solo units at slab widths and on the bench lattice, alternating in one box
out=gpurun_out/chain_ab.log; : > $out
for rep in 1 2; do
for a in "--nx 544" "--nx 544 --fuse-depth 3" "--nx 1056" "--nx 2080" "--nx 4096" "--nx 4096 --ny 2048 --dtype float64"; do
    for chain in 0 1; do
      echo -n "# $a chain $chain: " >> $out
      WT_CHAIN=$chain timeout -k 10 150 python3 bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 $a 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('units %5d columns/unit %3d depth %d %7.2f us/step %6.1f GLUPS'%(c['fuse_units'], c['fuse_chunk'], c['fuse_depth'], d['ms_per_step']*1e3, d['value']/1e3))
    else: print(l[:200])
" >> $out
    done
done
done
cat $out
