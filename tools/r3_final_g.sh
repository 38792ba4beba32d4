#!/bin/bash
# Runs ON THE GPU BOX: profiles and bench lines of the round-3 kernels with the MEASURED cut of the units (profiles/r03_g_*): kernel trace + FETCH_SIZE +
# WRITE_SIZE passes of the default bench, SQ counters of the bench kernel, the fp64 defaults, bench lines, the real slab splits (equal widths / cut by cost).
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3g; mkdir -p $out
# the profiler's per-kernel averages run over ALL dispatches: the trial passes of the measured cut (13 per mask, gather-path halo kernel) are kept out
# of them by WT_TUNE=0 (the modelled cut: same kernels, same bytes per pass, 1-2 % slower on this lattice); the bench lines below run the default
export WT_TUNE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/march4_trace -o t -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 > $out/march4_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/march4_fetch -o f -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --steps 48 --warmup 12 > $out/march4_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/march4_write -o w -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --steps 48 --warmup 12 > $out/march4_write.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/sq_$i -o c -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --steps 48 --warmup 12 > $out/sq_$i.log 2>&1 || echo "group $i failed"
done
for cfg in "4096 4096" "4096 2048"; do
  set -- $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/f64_$1x$2_trace -o t -- python3 bench.py --pmc-traffic 0 --dtype float64 --nx $1 --ny $2 --cpu-steps 0 > $out/f64_$1x$2_trace.log 2>&1
done
echo "profiles done"
unset WT_TUNE
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> /dev/null
python3 bench.py --dtype float64 --cpu-steps 0 > $out/bench_fp64.json 2> /dev/null
python3 bench.py --dtype float64 --nx 4096 --ny 2048 --cpu-steps 0 > $out/bench_fp64_cfg5.json 2> /dev/null
python3 bench.py --nx 1024 --ny 512 --steps 2000 --cpu-steps 0 > $out/bench_cfg2.json 2> /dev/null
python3 bench.py --local-slabs 8 --steps 408 --warmup 24 > $out/bench_local_slabs8.json 2> /dev/null
echo "bench lines done"
bash tools/r3_tune_ab.sh > $out/tune_ab.log 2>&1
find $PWD/gpurun_out -name "*.db" -size +30M -delete
tail -30 $out/tune_ab.log
