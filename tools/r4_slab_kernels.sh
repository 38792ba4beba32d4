#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of a plain and a body-holding slab of the real 8-way split (tools/r4_slab_kernels.py), per-kernel medians.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in ${@:-0 3}; do
  rocprofv3 --kernel-trace -d $R/gpurun_out/r4_q_prof_$r -o t -- python3 $R/tools/r4_slab_kernels.py $r 2>&1 | grep "slab"
  python3 - <<PY
import sqlite3, collections, glob
db = glob.glob("$R/gpurun_out/r4_q_prof_$r/**/*.db", recursive=True)[0]
c = sqlite3.connect(db); d = collections.defaultdict(list)
for name, dur in c.execute("select name, duration from kernels"):
    if "k_halo4" in name or "k_march3" in name or "k_step" in name: d[name.split("(")[0].replace("void wt::", "")].append(dur / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    print("   ", k, len(v), "calls: median %.2f min %.2f us" % (v[len(v) // 2], v[0]))
PY
done
