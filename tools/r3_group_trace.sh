# Runs ON THE GPU BOX: kernel traces of the locally linked 8-slab group, equal widths against the split cut by cost (per-kernel totals)
export TMPDIR=/tmp
out=$PWD/gpurun_out/grp; mkdir -p $out
for s in equal "cut by cost"; do
  tag=$(echo $s | tr ' ' '_')
  WT_SPLIT="$s" timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/$tag -o t -- python3 tools/r3_group_vs_alone.py > $out/$tag.log 2>&1
  grep "group wall" $out/$tag.log | cut -c1-200
  python3 tools/rocpd_summary.py $out/$tag/t_results.db | head -14
done
find $out -name "*.db" -size +30M -delete
