#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds (tools/ab/<name>.so; "" = the in-tree library) on a 544-column tunnel and the bench lattice:
# per-step time (wt_step_timed, 400 steps after 200) in alternating order, twice.   usage: tools/r4_ab.sh libA.so libB.so [widths...]
A=$1; B=$2; shift 2
W=${@:-"544 4096"}
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$A" "$B"; do
    echo "== ${lib:-in-tree} (run $rep)"
    WT_AB_LIB=$lib python3 $R/tools/r4_kernel_times.py $W 2>&1 | grep "us per step"
  done
done
