# Runs ON THE GPU BOX: the three numbers DESIGN §5 quotes, twice each, for box-to-box ranges (one call = one box)
for rep in 1 2; do for a in "--nx 544" "--nx 1056" "--nx 2080" ""; do
echo -n "${a:---nx 4096 (default)}: "; python3 bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.2f us/step %.1f GLUPS | contracted %.2f us/step | k_step %.2f us'%(d['ms_per_step']*1e3, d['value']/1e3, r.get('contracted',{}).get('ms_per_step',0)*1e3, r.get('single_step',{}).get('launch_ms',0)*1e3))"
done; done
