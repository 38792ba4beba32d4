"""Runs ON THE GPU BOX: what does the single step of a slab's refresh cycle cost on top of the fused passes?  A stand-alone 544-column handle stepped as
1 + 16 (what a slab with halo 17 does: the single k_step clears the seam buffer, the next pass builds its halo tables by the gather path) against 17 fused."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
ny = 4096
for nx, lo in ((544, 1700), (544, 100), (4096, 0)):
    mask = pkg.geometry.build_geometry(4096, ny, 10.0, None, "naca6409").mask[:, lo:lo + nx]
    mask = np.ascontiguousarray(mask)
    with pkg.Engine(nx, ny) as e:
        e.set_mask(mask); e.init_equilibrium(0.06); e.step(32, 0.58, 0.06)
        def cyc(parts, reps=24):
            e.sync(); tot = 0.0
            for _ in range(reps):
                for n in parts:
                    tot += e.step_timed(n, 0.58, 0.06)
            return tot / reps * 1e3
        a = min(cyc([16]) for _ in range(2)); b = min(cyc([1, 16]) for _ in range(2)); c = min(cyc([1]) for _ in range(2))
        print(f"{nx} columns at {lo}: 16 fused steps {a:.1f} us; 1 single + 16 fused {b:.1f} us (single step alone {c:.1f}); the cycle's surplus over 17 fused-rate steps: "
              f"{b - a * 17 / 16:.1f} us = {100 * (b / (a * 17 / 16) - 1):.1f} %", flush=True)
