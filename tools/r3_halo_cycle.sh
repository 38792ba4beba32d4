# Runs ON THE GPU BOX: the refresh cycle of a slab group on one GPU (bench.py --local-slabs 8, equal widths): halo 16 (1 + 4 + 4 + 4 + 3 steps) against 17 (1 + 4 x 4) and 13 / 21
for h in 16 17 13 21; do for rep in 1 2; do
  echo -n "halo $h: "; python3 bench.py --local-slabs 8 --halo $h --balance 0 --steps 408 --warmup 34 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('group wall %.2f us/step, sum of device %.2f, single steps %s' % (d['ms_per_step']*1e3, d['local_slabs']['sum_device_ms_per_step']*1e3, d['config']['single_steps'][:3]))"
done; done
