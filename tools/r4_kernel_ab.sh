#!/bin/bash
# Runs ON THE GPU BOX: k_halo4 / k_march3 durations (rocprofv3 kernel trace, same session) of builds of the library (tools/ab/<name>.so; "" = in-tree)
# on a 544-column tunnel and on the bench lattice.   usage: tools/r4_kernel_ab.sh libA.so libB.so ...
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  tag=${lib:-current}
  rm -rf $R/gpurun_out/r4_kab_$tag
  WT_AB_LIB=$lib rocprofv3 --kernel-trace -d $R/gpurun_out/r4_kab_$tag -o t -- python3 $R/tools/r4_kernel_times.py 544 4096 2>&1 | grep "us per step"
  python3 - <<PY
import sqlite3, collections, glob
db = glob.glob("$R/gpurun_out/r4_kab_$tag/**/*.db", recursive=True)[0]
c = sqlite3.connect(db); d = collections.defaultdict(list)
for name, dur, gx in c.execute("select name, duration, grid_x from kernels"):
    if "k_halo4" in name or "k_march3" in name: d[(name.split("(")[0].replace("void wt::", ""), gx)].append(dur / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    if len(v) > 20: print("  $tag", k, len(v), "calls: avg %.2f median %.2f min %.2f us" % (sum(v) / len(v), v[len(v) // 2], v[0]))
PY
  rm -rf $R/gpurun_out/r4_kab_$tag
done
done
