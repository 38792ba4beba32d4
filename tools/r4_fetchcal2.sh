#!/bin/bash
# Runs ON THE GPU BOX: the request counters of tools/kfetchcal's kernels only (the gather shape of k_halo4 added).  Output: gpurun_out/r4u2/cal_summary.txt
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4u2; mkdir -p $out
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $grp | tr ' ' '+' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/cal_$tag -o c -- ./tools/kfetchcal 2 > $out/cal_$tag.log 2>&1 || echo "cal $tag failed" >> $out/failed.txt
done
for d in $out/cal_* ; do [ -d $d ] && { echo "== $(basename $d)"; python3 tools/sq_summary.py "read_gather" $(find $d -name "*.db"); python3 tools/sq_summary.py "read8" $(find $d -name "*.db"); } ; done > $out/cal_summary.txt 2>&1
cat $out/cal_summary.txt
