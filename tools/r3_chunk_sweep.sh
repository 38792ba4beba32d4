#!/bin/bash
# Runs ON THE GPU BOX: unit length (fuse_chunk) x steps per pass with chain blocks, slab widths and the bench lattice
out=gpurun_out/chunk_sweep.log; : > $out
run() { echo "# $*" >> $out; timeout -k 10 150 python3 bench.py --ny 4096 --fuse 2 --cpu-steps 0 --steps 408 --warmup 24 "$@" >> $out 2>&1; }
for depth in 3 4; do
  for c in 0 4 5 6 7; do run --nx 544 --fuse-depth $depth --fuse-chunk $c; done
  for c in 0 6 8 9 11 13; do run --nx 1056 --fuse-depth $depth --fuse-chunk $c; done
  for c in 0 9 12 17 23; do run --nx 4096 --fuse-depth $depth --fuse-chunk $c; done
done
