#!/bin/bash
# Runs ON THE GPU BOX (round 3, first batch): (A) resident waves per SIMD x steps per pass at slab widths, (B) kernel split of the
# slab-sized pass, (C) SQ / TCP counters of the kernel that prints the bench line, (D) FETCH/WRITE traffic of the fp64 defaults.
# Every rocprofv3 command has the program itself behind `--`; --pmc passes carry no trace domain besides the kernel trace.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3b1
mkdir -p $out
: > $out/waves.log
for w in 544 1056; do
  for waves in 1 2 3; do
    for depth in 2 3; do
      echo "# width $w waves $waves depth $depth" >> $out/waves.log
      WT_MARCH_WAVES=$waves timeout -k 10 150 python3 bench.py --nx $w --ny 4096 --fuse 2 --fuse-depth $depth --cpu-steps 0 --steps 402 >> $out/waves.log 2>&1
    done
  done
done
echo "A done"
for w in 544 1056; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/trace_$w -o t -- python3 bench.py --pmc-traffic 0 --nx $w --ny 4096 --cpu-steps 0 --steps 402 > $out/trace_$w.log 2>&1
done
echo "B done"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_WRITE_REQ TCP_TCP_TA_DATA_STALL_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/sq_$i -o c -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --steps 48 --warmup 12 > $out/sq_$i.log 2>&1 || echo "group $i failed"
done
echo "C done"
for cfg in "4096 4096" "4096 2048"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c -d $out/f64_$1x$2_$c -o c -- python3 bench.py --pmc-traffic 0 --dtype float64 --nx $1 --ny $2 --cpu-steps 0 --steps 48 --warmup 12 > $out/f64_$1x$2_$c.log 2>&1 || echo "f64 $cfg $c failed"
  done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/f64_$1x$2_trace -o t -- python3 bench.py --pmc-traffic 0 --dtype float64 --nx $1 --ny $2 --cpu-steps 0 > $out/f64_$1x$2_trace.log 2>&1
done
echo "D done"
find $out -name "*.db" -size +30M -delete
du -sh $out
