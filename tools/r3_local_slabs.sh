#!/bin/bash
# Runs ON THE GPU BOX: the column-slab path with all slabs on ONE GPU (in-process transport) — bench lines for P = 1, 2, 4, 8 and a
# kernel trace of the 8-slab run (the in-process transport moves ghost columns with a copy kernel) (is the ghost refresh on the comm stream inside the interior kernel's interval?)
export TMPDIR=/tmp
out=gpurun_out/local_slabs; mkdir -p $out
python3 bench.py --cpu-steps 0 --steps 340 --warmup 34 > $out/p1.json 2> $out/p1.err
for P in 2 4 8; do
  python3 bench.py --local-slabs $P --halo 17 --steps 340 --warmup 34 > $out/p$P.json 2> $out/p$P.err
done
python3 bench.py --local-slabs 8 --halo 1 --fuse 0 --steps 100 --warmup 10 > $out/p8_halo1.json 2> $out/p8_halo1.err
timeout -k 10 120 rocprofv3 --kernel-trace -d $out/trace8 -o t -- python3 bench.py --pmc-traffic 0 --local-slabs 8 --halo 17 --steps 68 --warmup 17 > $out/trace8.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace -d $out/trace8h1 -o t -- python3 bench.py --pmc-traffic 0 --local-slabs 8 --halo 1 --fuse 0 --steps 20 --warmup 5 > $out/trace8h1.log 2>&1
tail -c 600 $out/p8.json; ls -la $out/trace8 $out/trace8h1
timeout -k 10 120 rocprofv3 --kernel-trace -d $out/trace2 -o t -- python3 bench.py --pmc-traffic 0 --local-slabs 2 --halo 17 --steps 136 --warmup 17 > $out/trace2.log 2>&1
