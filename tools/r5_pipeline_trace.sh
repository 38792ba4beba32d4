#!/bin/bash
# Runs ON THE GPU BOX: timeline of the last pipelined passes (start / end of every kernel relative to the first, by stream).  usage: tools/r5_pipeline_trace.sh K
export TMPDIR=/tmp
K=${1:-4}
out=$PWD/gpurun_out/r5_pipe_trace_$K; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o t -- python3 tools/r5_pipeline_trace.py $K > $out/run.log 2>&1
python3 - <<PY
import sqlite3, glob
db = glob.glob("$out/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute(f"select name, start, end, {qcol or 0}, grid_x from kernels order by start"))
rows = [r for r in rows if "k_march3" in r[0] or "k_halo" in r[0]]
last = rows[-(2 * $K * 3):] if $K > 1 else rows[-6:]
t0 = last[0][1]
for n, s, e, q, g in last:
    nm = "halo " if "k_halo" in n else "march"
    print(f"{nm} queue {q} grid {g:6d}: start {(s - t0) / 1e3:8.1f} us  end {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:6.1f} us)")
PY
rm -rf $out
