"""Runs ON THE GPU BOX: the slabs of the 8-way split of the bench lattice (tools/r3_slab_costs.py) at forced steps per pass 2 / 3 / 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096; halo = 16; P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
for depth in (2, 3, 4):
    costs = []
    for r in range(P):
        x0, x1 = r * nx // P, (r + 1) * nx // P
        lo, hi = max(0, x0 - halo), min(nx, x1 + halo)
        sub = np.ascontiguousarray(mask[:, lo:hi])
        with pkg.Engine(hi - lo, ny) as e:
            e.set_option("fuse_depth", depth)
            e.set_mask(sub); e.init_equilibrium(0.06); e.step(24, 0.58, 0.06)
            costs.append(e.step_timed(408, 0.58, 0.06) / 408 * 1e3)
    print(f"N = {P}, {depth} steps per pass: " + "  ".join(f"{c:.2f}" for c in costs) + f"   slowest {max(costs):.2f}  sum {sum(costs):.1f}")
