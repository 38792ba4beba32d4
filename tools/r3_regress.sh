for rep in 1 2; do for lib in lib_ebbe5b9.so lib_head.so lib_head2.so; do for a in "--nx 544" "--nx 1056"; do
echo -n "$lib $a: "; WT_AB_LIB=$lib python3 tools/ab_bench.py --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f us/step units %d'%(d['ms_per_step']*1e3, d['config']['fuse_units']))"
done; done; done
