#!/bin/bash
# Runs ON THE GPU BOX: what do the SQ counters behind bench.py's valu_busy_frac read on kernels that are VALU-saturated by construction (tools/kvalu)?
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4x; mkdir -p $out
./tools/kvalu > $out/kvalu_times.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY -d $out/valu -o c -- ./tools/kvalu > $out/valu.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace -d $out/trace -o t -- ./tools/kvalu > $out/trace.log 2>&1
python3 - <<'PY' > $out/valu_summary.txt 2>&1
import sqlite3, glob, os
from collections import defaultdict
out = os.path.join(os.environ.get("PWD", "."), "gpurun_out/r4x")
db = glob.glob(out + "/valu/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = defaultdict(dict)
for kname, cname, disp, val in c.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection"):
    rows[(disp, kname)][cname] = rows[(disp, kname)].get(cname, 0.0) + val
for (disp, k), r in sorted(rows.items()):
    gui = r.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if gui < 1e5: continue            # the warm launches
    print(f"dispatch {disp:3d} {k[:40]:40s} INSTS_VALU/SIMD {r['SQ_INSTS_VALU']/1024:10.0f}  ACTIVE_VALU x4/SIMD {4*r['SQ_ACTIVE_INST_VALU']/1024:10.0f}  cycles {gui:10.0f}"
          f"  busy = {4*r['SQ_ACTIVE_INST_VALU']/1024/gui:.3f}   cycles per instr = {gui/(r['SQ_INSTS_VALU']/1024):.2f}   ACTIVE_ANY x4/SIMD/cycles {4*r.get('SQ_ACTIVE_INST_ANY',0)/1024/gui:.3f}")
PY
cat $out/kvalu_times.txt; cat $out/valu_summary.txt
