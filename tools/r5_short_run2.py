"""Developer experiment: device time per pass over consecutive 20-step regions on a FRESH handle (as bench.py --side 0 sees it), then after a re-initialisation.
    python3 tools/r5_short_run2.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg
mask = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
with pkg.Engine(4096, 4096) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(5, 0.58, 0.06); e.sync()
    ts = [e.step_timed(20, 0.58, 0.06) / 5 * 1e3 for _ in range(24)]
    print("fresh handle, init + 5 steps: us per pass over consecutive 20-step regions:\n   " + " ".join(f"{t:.0f}" for t in ts), flush=True)
    print("   clamp events:", e.clamp_events(), "fast_div_active", e.get_option("fast_div_active"), "two_op", e.get_option("fast_div_two_op_active"))
    e.init_equilibrium(0.06); e.step(5, 0.58, 0.06); e.sync()
    ts = [e.step_timed(20, 0.58, 0.06) / 5 * 1e3 for _ in range(24)]
    print("same handle, init + 5 steps again:\n   " + " ".join(f"{t:.0f}" for t in ts), flush=True)
    e.init_equilibrium(0.06); e.step(5, 0.58, 0.06); e.sync()
    ts = []
    for _ in range(12):
        ts.append(e.step_timed(20, 0.58, 0.06) / 5 * 1e3)
        ev = e.clamp_events()
    print("same, reading clamp_events between the regions:\n   " + " ".join(f"{t:.0f}" for t in ts), ev, flush=True)
