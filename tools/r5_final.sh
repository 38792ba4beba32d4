#!/bin/bash
# Runs ON THE GPU BOX: the profiles and bench lines of round 5's final library (profiles/r05_w_*), one session: kernel trace + FETCH_SIZE + WRITE_SIZE passes of the
# default bench, its timed region, the SQ counters of round 4's library and this one side by side, bench lines of every BASELINE configuration, the driver's
# arguments, a local 8-slab group, the 4-rank line over the RCCL stand-in.      usage: bash tools/r5_final.sh [part ...]   (parts: prof sq bench slabs; default all)
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r5w; mkdir -p $out
parts=${@:-"prof sq bench slabs"}
for part in $parts; do
case $part in
prof)
  # the profiler's per-kernel averages run over ALL dispatches: the trial passes of the measured cut are kept out of them by WT_TUNE=0 (as in round 4)
  export WT_TUNE=0
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/march4_trace -o t -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 > $out/march4_trace.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/march4_fetch -o f -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --steps 48 --warmup 12 > $out/march4_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/march4_write -o w -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --steps 48 --warmup 12 > $out/march4_write.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/cfg4_trace -o t -- python3 bench.py --config 4 --pmc-traffic 0 --cpu-steps 0 > $out/cfg4_trace.log 2>&1
  unset WT_TUNE
  bash tools/r4_trace_timed.sh > $out/timed_region.log 2>&1; cp gpurun_out/r4w4/timed_region.txt $out/timed_region.txt 2>/dev/null
  echo "profiles done";;
sq)
  # vector-instruction activity per launch of the bench kernel: round 4's library against this one, same session
  for lib in lib_r4.so ""; do
    tag=${lib:-current}
    WT_TUNE=0 WT_AB_LIB=$lib timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/sq_$tag -o s -- python3 tools/ab_bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --side 0 --steps 48 --warmup 12 > $out/sq_$tag.log 2>&1
    python3 tools/sq_summary.py k_march3 $(find $out/sq_$tag -name "*.db") > $out/sq_$tag.txt 2>&1
  done
  echo "sq done";;
bench)
  python3 bench.py > $out/bench.json 2> $out/bench.err
  python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> /dev/null
  python3 bench.py --steps 20 --warmup 5 --side 0 > $out/bench_driver_args_no_preheat.json 2> /dev/null
  for c in 0 1 3 4; do python3 bench.py --config $c --cpu-steps 3 > $out/bench_cfg$c.json 2> /dev/null; done
  echo "bench lines done";;
slabs)
  python3 bench.py --local-slabs 8 --steps 610 --warmup 61 > $out/bench_local_slabs8.json 2> /dev/null
  g++ -O2 -std=c++17 -fPIC -shared -o /tmp/librccl_stub.so $R/tests/_rccl_stub/rccl_stub.cpp -ldl -lrt -lpthread && \
    LD_PRELOAD=/tmp/librccl_stub.so WT_BENCH_FORCE_DEVICE=0 WT_BENCH_TORCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout -k 10 500 python3 bench.py --gpus 4 --cpu-steps 0 --steps 122 --warmup 61 > $out/bench_4ranks_stub.json 2> $out/bench_4ranks_stub.err
  echo "slab lines done";;
esac
done
find $R/gpurun_out -name "*.db" -size +30M -delete
ls $out
