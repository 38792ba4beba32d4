# Runs ON THE GPU BOX: one-GPU bench.py at slab widths x {single steps, two steps per pass, three steps per pass}, one JSON
# line each into gpurun_out/width_sweep.log
set -e
out=gpurun_out/width_sweep.log; : > $out
for w in 288 544 1056 2080 4096; do
  timeout -k 10 200 python bench.py --nx $w --ny 4096 --fuse 0 --cpu-steps 0 --steps 402 >> $out 2>&1
  timeout -k 10 200 python bench.py --nx $w --ny 4096 --fuse 2 --fuse-depth 2 --cpu-steps 0 --steps 402 >> $out 2>&1
  timeout -k 10 200 python bench.py --nx $w --ny 4096 --fuse 2 --fuse-depth 3 --cpu-steps 0 --steps 402 >> $out 2>&1
done
