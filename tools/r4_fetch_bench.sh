#!/bin/bash
# Runs ON THE GPU BOX: the L2 / L1 request counters of the bench kernels only (see tools/r4_fetchcal.sh).  usage: tools/r4_fetch_bench.sh <tag>
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4w_$1; mkdir -p $out
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '+' | cut -c1-40)
  WT_TUNE=0 timeout -k 10 300 rocprofv3 --pmc $grp -d $out/bench_$tag -o c -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --side 0 --steps 24 --warmup 8 > $out/bench_$tag.log 2>&1 || echo "bench $tag failed" >> $out/failed.txt
done
WT_TUNE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --side 0 --steps 200 --warmup 40 > $out/trace.log 2>&1
python3 tools/rocpd_summary.py $(find $out/trace -name "*.db") > $out/trace_summary.txt 2>&1
for d in $out/bench_* ; do [ -d $d ] && { echo "== $d"; python3 tools/sq_summary.py "k_" $(find $d -name "*.db"); } ; done 2>&1 | grep -v "rocclr\|k_verify\|k_classify\|k_bounce\|k_seam\|k_fill\|k_mask" | sed 's#/tmp/code/[^ ]*/gpurun_out/##' > $out/bench_summary.txt
find $out -name "*.db" -size +20M -delete
cat $out/bench_summary.txt; grep "k_march3\|k_halo" $out/trace_summary.txt
