# Runs ON THE GPU BOX: slabs cut by measured cost (bench.py --local-slabs P, --balance R) against equal widths
for P in 8 4 2; do
  echo "== $P slabs"
  python3 bench.py --local-slabs $P --steps 408 --warmup 24 --balance 4 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('edges', c['edges'])
for h in c['balance']: print('  widths', h['widths'], 'us/step', h['slab_us_per_step'], 'slowest', h['slowest'])
print('  group wall %.2f us/step (all slabs on ONE GPU), sum of device %.2f' % (d['ms_per_step']*1e3, d['local_slabs']['sum_device_ms_per_step']*1e3))"
done
