"""Runs ON THE GPU BOX: two against four steps per pass on slab-sized handles of the bench tunnel (plain columns / the columns over the thick part
of the body), local widths 300 ... 560: where does the automatic choice have to flip?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096
dtype = os.environ.get("DTYPE", "float32")
nwin = ny // (128 if dtype == "float32" else 64)
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
for w in [int(a) for a in sys.argv[1:]] or (300, 340, 380, 420, 460, 500, 560):
    row = []
    for name, lo in (("plain", 100), ("body", 1700)):
        sub = np.ascontiguousarray(mask[:, lo:lo + w])
        for depth in (0, 2, 3, 4):
            with pkg.Engine(w, ny, dtype=dtype) as e:
                e.set_option("fuse_depth", depth) if depth else e.set_option("fuse_steps", 0)
                e.set_mask(sub); e.init_equilibrium(0.06); e.step(24, 0.58, 0.06)
                us = min(e.step_timed(408, 0.58, 0.06) for _ in range(2)) / 408 * 1e3
                row.append(f"{name} {depth}/pass {us:.2f}")
    print(f"{w} local columns ({(w - 4) * nwin // 2048} per unit, {dtype}): " + "   ".join(row), flush=True)
