"""Developer experiment: why does a timed region of 20 steps read 7 % above the steady state?  Consecutive 20-step timings after init_equilibrium + 5 steps,
with and without 400 steps of the same handle in front, device time per four-step pass.
    python3 tools/r5_short_run.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg
mask = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
with pkg.Engine(4096, 4096) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(400, 0.58, 0.06); e.sync()
    for label, pre in (("init + 5 steps", 5), ("init + 5 steps", 5), ("init + 105 steps", 105), ("init + 405 steps", 405)):
        e.init_equilibrium(0.06); e.step(pre, 0.58, 0.06); e.sync()
        ts = [e.step_timed(20, 0.58, 0.06) / 5 * 1e3 for _ in range(8)]
        print(f"{label}: us per pass over consecutive 20-step regions: " + " ".join(f"{t:.1f}" for t in ts), flush=True)
    # the same regions with a pause of 50 ms in front of each
    e.init_equilibrium(0.06); e.step(405, 0.58, 0.06); e.sync()
    ts = []
    for _ in range(5):
        time.sleep(0.05)
        ts.append(e.step_timed(20, 0.58, 0.06) / 5 * 1e3)
    print("after 405 steps, 50 ms of idle before every region: " + " ".join(f"{t:.1f}" for t in ts), flush=True)
    ev = e.clamp_events()
    print("clamp events now:", ev)
