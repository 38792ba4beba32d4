#!/bin/bash
# Runs ON THE GPU BOX: round-5 A/B of library builds (tools/ab/<name>.so; "" = the in-tree library): per-step time of a 544-column slab-sized tunnel and the
# bench lattice (fp32, alternating, twice) and of BASELINE configs[4] (fp64).   usage: tools/r5_ab.sh libA.so ""
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== ${lib:-in-tree} (run $rep)"
    WT_AB_LIB=$lib python3 $R/tools/r4_kernel_times.py 544 4096 2>&1 | grep "us per step"
    WT_AB_LIB=$lib python3 $R/tools/ab_bench.py --config 4 --cpu-steps 0 --pmc-traffic 0 --side 0 --steps 200 --warmup 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('  fp64 cfg4', round(d['value']), 'MLUPS', round(d['ms_per_step']*1e3,2), 'us/step')"
  done
done
