#!/bin/bash
# Runs ON THE GPU BOX: fp64 (BASELINE configs[4]) under two builds of the library, kernel trace of each.   usage: tools/r4_f64_ab.sh libA.so ""
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  tag=${lib:-current}
  rm -rf $R/gpurun_out/r4_f64_$tag
  WT_AB_LIB=$lib rocprofv3 --kernel-trace -d $R/gpurun_out/r4_f64_$tag -o t -- python3 $R/tools/ab_bench.py --config 4 --cpu-steps 0 --pmc-traffic 0 --side 0 --steps 200 --warmup 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('  $tag', round(d['value']), 'MLUPS', round(d['ms_per_step']*1e3,2), 'us/step')"
  python3 - <<PY
import sqlite3, collections, glob
db = glob.glob("$R/gpurun_out/r4_f64_$tag/**/*.db", recursive=True)[0]
c = sqlite3.connect(db); d = collections.defaultdict(list)
for name, dur, gx in c.execute("select name, duration, grid_x from kernels"):
    if "k_halo" in name or "k_march3" in name: d[(name.split("(")[0].replace("void wt::", ""), gx)].append(dur / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    if len(v) > 20: print("  $tag", k, len(v), "calls: avg %.2f median %.2f min %.2f us" % (sum(v) / len(v), v[len(v) // 2], v[0]))
PY
  rm -rf $R/gpurun_out/r4_f64_$tag
done
done
