# Runs ON THE GPU BOX: one-GPU bench.py at slab widths / sites per lane / fp64, one JSON line each into gpurun_out/width_sweep.log
set -e
out=gpurun_out/width_sweep.log; : > $out
for w in 288 544 1056; do
  for wv in 2 3; do WT_MARCH_WAVES=$wv timeout -k 10 200 python bench.py --nx $w --ny 4096 --fuse 2 --fuse-sites 2 --cpu-steps 0 --steps 400 >> $out 2>&1; done
done
for w in 288 544 1056 2080 4096; do timeout -k 10 200 python bench.py --nx $w --ny 4096 --cpu-steps 0 --steps 400 >> $out 2>&1; done
