#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE and the L2's raw request counters for (1) tools/kfetchcal (known byte counts in the marching kernels'
# access shapes) and (2) the bench kernel itself.  Output: gpurun_out/r4u/*.txt
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4u; mkdir -p $out
rocprofv3 -L > $out/counters_avail.txt 2>&1
./tools/kfetchcal 5 > $out/kfetchcal_times.txt 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $grp | tr ' ' '+' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/cal_$tag -o c -- ./tools/kfetchcal 2 > $out/cal_$tag.log 2>&1 || echo "cal $tag failed" >> $out/failed.txt
  WT_TUNE=0 timeout -k 10 300 rocprofv3 --pmc $grp -d $out/bench_$tag -o c -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --side 0 --steps 24 --warmup 8 > $out/bench_$tag.log 2>&1 || echo "bench $tag failed" >> $out/failed.txt
done
for d in $out/cal_* ; do [ -d $d ] && { echo "== $d"; python3 tools/sq_summary.py "" $(find $d -name "*.db"); } ; done > $out/cal_summary.txt 2>&1
for d in $out/bench_* ; do [ -d $d ] && { echo "== $d"; python3 tools/sq_summary.py "k_" $(find $d -name "*.db"); } ; done > $out/bench_summary.txt 2>&1
find $PWD/gpurun_out/r4u -name "*.db" -size +20M -delete
tail -5 $out/kfetchcal_times.txt
