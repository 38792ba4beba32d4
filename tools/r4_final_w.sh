#!/bin/bash
# Runs ON THE GPU BOX: the profiles and bench lines of round 4's final library (profiles/r04_w_*): kernel trace + FETCH_SIZE + WRITE_SIZE passes of the
# default bench, bench lines of every BASELINE configuration, the driver's arguments, a local 8-slab group, the frame loop.
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4w; mkdir -p $out
# the profiler's per-kernel averages run over ALL dispatches: the trial passes of the measured cut (13 per mask, gather-path halo kernel) are kept out
# of them by WT_TUNE=0 (the modelled cut: same kernels, same bytes per pass, 1-2 % slower on this lattice); the bench lines below run the default
export WT_TUNE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/march4_trace -o t -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 > $out/march4_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/march4_fetch -o f -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --steps 48 --warmup 12 > $out/march4_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/march4_write -o w -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --steps 48 --warmup 12 > $out/march4_write.log 2>&1
echo "profiles done"
unset WT_TUNE
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --side 0 > $out/bench_driver_args_no_preheat.json 2> /dev/null
for c in 0 1 3 4; do python3 bench.py --config $c --cpu-steps 3 > $out/bench_cfg$c.json 2> /dev/null; done
python3 bench.py --local-slabs 8 --steps 580 --warmup 58 > $out/bench_local_slabs8.json 2> /dev/null
python3 tools/r4_frame_loop.py 40 > $out/frame_loop.txt 2>&1
echo "bench lines done"
find $PWD/gpurun_out -name "*.db" -size +30M -delete
ls $out
