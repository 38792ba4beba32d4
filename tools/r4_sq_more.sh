#!/bin/bash
# Runs ON THE GPU BOX: the SQ counters that say where a marching wave's non-arithmetic time goes (vector-memory issue stalls, LDS, scalar), bench kernel.
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4sq; mkdir -p $out
i=0
for grp in "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  WT_TUNE=0 timeout -k 10 300 rocprofv3 --pmc $grp -d $out/g$i -o c -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 --side 0 --steps 24 --warmup 8 > $out/g$i.log 2>&1 || echo "group $i failed" >> $out/failed.txt
done
for d in $out/g*; do [ -d $d ] && python3 tools/sq_summary.py "k_march3<float, 2, 4, false" $(find $d -name "*.db"); done > $out/summary.txt 2>&1
cat $out/summary.txt; cat $out/failed.txt 2>/dev/null
find $out -name "*.db" -size +20M -delete
