# Runs ON THE GPU BOX: the longest chain unit (WT_MAX_CHAIN, default 160) on lattices whose one resident round asks for longer units
for cfg in "8192 4096 float32" "16384 4096 float32" "8192 4096 float64"; do set -- $cfg
  for mc in 160 320; do for rep in 1 2; do
    echo -n "$1 x $2 $3, max chain $mc: "; WT_MAX_CHAIN=$mc python3 bench.py --nx $1 --ny $2 --dtype $3 --cpu-steps 0 --steps 204 --warmup 24 --pmc-traffic 0 --fast-math 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f us/step %.1f GLUPS units %d chunk %d'%(d['ms_per_step']*1e3, d['value']/1e3, d['config']['fuse_units'], d['config']['fuse_chunk']))"
  done; done
done
