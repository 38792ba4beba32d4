#!/bin/bash
# Runs ON THE GPU BOX (round 3, profiles of the final kernels): kernel trace + FETCH_SIZE + WRITE_SIZE passes of bench.py for the default (four
# steps per pass, chain blocks), three steps, two steps and single steps; the fp64 defaults; SQ / TCP counters of the bench kernel; plain bench lines.
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3b2; mkdir -p $out
bash tools/profile_bench.sh r03 > $out/profile_bench.log 2>&1
for cfg in "4096 4096" "4096 2048"; do
  set -- $cfg
  tag=f64_$1x$2
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t -- python3 bench.py --pmc-traffic 0 --dtype float64 --nx $1 --ny $2 --cpu-steps 0 > $out/${tag}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c -d $out/${tag}_$c -o c -- python3 bench.py --pmc-traffic 0 --dtype float64 --nx $1 --ny $2 --cpu-steps 0 --steps 48 --warmup 12 > $out/${tag}_$c.log 2>&1 || echo "$tag $c failed"
  done
done
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_WRITE_REQ TCP_TCP_TA_DATA_STALL_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/sq_$i -o c -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --steps 48 --warmup 12 > $out/sq_$i.log 2>&1 || echo "group $i failed"
done
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> /dev/null
python3 bench.py --dtype float64 --cpu-steps 0 > $out/bench_fp64.json 2> /dev/null
python3 bench.py --dtype float64 --nx 4096 --ny 2048 --cpu-steps 0 > $out/bench_fp64_cfg5.json 2> /dev/null
python3 bench.py --fuse 0 --cpu-steps 0 > $out/bench_nofuse.json 2> /dev/null
find $PWD/gpurun_out -name "*.db" -size +30M -delete
du -sh $out gpurun_out/prof_r03
