"""Runs ON THE GPU BOX: what would each slab of an N-way split of the bench lattice cost per step?  For every rank r a stand-alone handle of the
slab's local width (owned + ghost columns) is stepped on the mask columns that slab holds (so the body pieces are the real ones; the handle's
own edges are treated as inlet / outlet, which a slab's are not — a small pessimism), 408 timed steps each.  The slowest slab sets the pace of
the N-GPU run; this is still a one-GPU PROJECTION (no exchange, no refresh steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg

nx = ny = 4096
halo = 16
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(24, 0.58, 0.06)
    whole = e.step_timed(408, 0.58, 0.06) / 408 * 1e3
    gain = e.get_option("tune_gain")
print(f"whole 4096^2 lattice on one GPU: {whole:.2f} us/step = {nx * ny / whole / 1e3:.1f} GLUPS (tune gain {gain:.3f})")
for P in (2, 4, 8):
    costs = []
    for r in range(P):
        x0, x1 = r * nx // P, (r + 1) * nx // P
        lo, hi = max(0, x0 - halo), min(nx, x1 + halo)
        sub = np.ascontiguousarray(mask[:, lo:hi])
        with pkg.Engine(hi - lo, ny) as e:
            e.set_mask(sub); e.init_equilibrium(0.06); e.step(24, 0.58, 0.06)
            us = e.step_timed(408, 0.58, 0.06) / 408 * 1e3
            costs.append((us, int(sub.any()), int(e.get_option("fuse_depth")), int(e.get_option("chain_units")), int(e.get_option("fuse_units")), e.get_option("tune_gain")))
    worst = max(c[0] for c in costs)
    print(f"N = {P}: per-slab us/step " + "  ".join(f"{c[0]:.2f}{'*' if c[1] else ''}" for c in costs) + f"   (* holds a piece of the body)")
    print(f"        slowest {worst:.2f} us/step -> {nx * ny / worst / 1e3:.0f} GLUPS if the exchange hides = {whole / worst:.2f} x the one-GPU run; depth {costs[0][2]}, chain units {[c[3] for c in costs]}, tune gains {[round(c[5], 3) for c in costs]}")
