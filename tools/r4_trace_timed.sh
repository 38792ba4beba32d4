#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of the DEFAULT bench (measured cut of the units, as the bench line is produced), read over the timed region only —
# the last `steps / 4` passes — so that the trial passes of the measured cut and the warm-up stay out of the averages.  Output: gpurun_out/r4w4/timed_region.txt
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4w4; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --pmc-traffic 0 --cpu-steps 0 --fast-math 0 > $out/trace.log 2>&1
python3 - <<PY > $out/timed_region.txt
import sqlite3, glob, json
db = glob.glob("$out/trace/**/*.db", recursive=True)[0]
line = [l for l in open("$out/trace.log") if l.startswith("{")][-1]
d = json.loads(line)
passes = d["steps"] // d["config"]["pass_depth"]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
march = [(s, e) for n, s, e in rows if "k_march3" in n and ",false," in n.replace(" ", "")]
halo = [(s, e) for n, s, e in rows if "k_halo4" in n or "k_halo3" in n]
m, h = march[-passes:], halo[-passes:]
am, ah = sum(e - s for s, e in m) / len(m) / 1e3, sum(e - s for s, e in h) / len(h) / 1e3
span = (m[-1][1] - h[0][0]) / 1e3 / passes
print(f"bench.py under rocprofv3 --kernel-trace (default settings: measured cut), the last {passes} passes = the timed region of {d['steps']} steps:")
print(f"  k_march3 (non-emitting) {len(m)} launches: average {am:.1f} us   k_halo4 {len(h)} launches: average {ah:.1f} us   sum {am + ah:.1f} us per pass")
print(f"  first halo kernel's start to last marching kernel's end / passes: {span:.1f} us per pass (with the gaps between launches)")
print(f"  the same run's bench line: launch_ms {d['roofline']['launch_ms'] * 1e3:.1f} us per pass (HIP events on the library's stream), {d['value']:.0f} MLUPS under the profiler")
PY
cat $out/timed_region.txt
find $out -name "*.db" -size +30M -delete
