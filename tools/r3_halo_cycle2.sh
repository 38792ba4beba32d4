# Runs ON THE GPU BOX: equal widths against slabs cut by cost, all 8 slabs on ONE GPU (where only the SUM of the slabs' work counts)
for b in 0 3 0 3; do
  echo -n "balance $b: "; python3 bench.py --local-slabs 8 --halo 16 --balance $b --steps 408 --warmup 34 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('group wall %.2f us/step, sum of device %.2f, widths %s, alone %s, depth %s units %s' % (d['ms_per_step']*1e3, d['local_slabs']['sum_device_ms_per_step']*1e3, [b-a for a,b in zip(c['edges'][:-1],c['edges'][1:])], (c['balance'][-1]['slab_us_per_step'] if c['balance'] else None), c['fuse_depth'][:2], c.get('fuse_units')))"
done
