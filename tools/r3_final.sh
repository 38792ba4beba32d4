#!/bin/bash
# Runs ON THE GPU BOX: the profiles and bench lines of the FINAL round-3 kernels (profiles/r03_e_*): kernel trace + FETCH_SIZE + WRITE_SIZE passes of
# bench.py (default, three / two steps per pass, single steps), the fp64 defaults, the slab-width sweep with the automatic plan, plain bench lines.
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3e; mkdir -p $out
bash tools/profile_bench.sh r03e --fast-math 0 > $out/profile_bench.log 2>&1
for cfg in "4096 4096" "4096 2048"; do
  set -- $cfg
  tag=f64_$1x$2
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t -- python3 bench.py --pmc-traffic 0 --dtype float64 --nx $1 --ny $2 --cpu-steps 0 > $out/${tag}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c -d $out/${tag}_$c -o c -- python3 bench.py --pmc-traffic 0 --dtype float64 --nx $1 --ny $2 --cpu-steps 0 --steps 48 --warmup 12 > $out/${tag}_$c.log 2>&1 || echo "$tag $c failed"
  done
done
: > $out/width_sweep.txt
for w in 288 416 544 800 1056 2080 3072 4096; do
  for rep in 1 2; do
    echo -n "$w x 4096 fp32, automatic plan: " >> $out/width_sweep.txt
    python3 bench.py --nx $w --ny 4096 --cpu-steps 0 --steps 408 --warmup 24 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('steps/pass %d units %5d (chain %s) columns/unit %3d  %7.2f us/step %6.1f GLUPS | contracted %.2f us/step | k_step %.2f us/step'%(c['fuse_depth'], c['fuse_units'], '-', c['fuse_chunk'], d['ms_per_step']*1e3, d['value']/1e3, r.get('contracted',{}).get('ms_per_step',0)*1e3, r.get('single_step',{}).get('launch_ms',0)*1e3))" >> $out/width_sweep.txt
  done
done
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> /dev/null
python3 bench.py --dtype float64 --cpu-steps 0 > $out/bench_fp64.json 2> /dev/null
python3 bench.py --dtype float64 --nx 4096 --ny 2048 --cpu-steps 0 > $out/bench_fp64_cfg5.json 2> /dev/null
python3 bench.py --fuse 0 --cpu-steps 0 > $out/bench_nofuse.json 2> /dev/null
python3 bench.py --nx 1024 --ny 512 --steps 2000 --cpu-steps 0 > $out/bench_cfg2.json 2> /dev/null
find $PWD/gpurun_out -name "*.db" -size +30M -delete
cat $out/width_sweep.txt
