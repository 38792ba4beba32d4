"""Developer experiment: at which whole-lattice sizes do overlapping windows pay?  (cache-resident lattices are not bound by line traffic)
    python3 tools/r5_ovl_sizes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg
for nx, ny, shape, aoa in ((512, 256, "naca2412", 5.0), (1024, 512, "naca2412", 5.0), (2048, 1024, "naca2412", 5.0), (2048, 2048, "naca6409", 10.0), (4096, 2048, "naca6409", 10.0), (3072, 3072, "naca6409", 10.0)):
    mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
    res = {}
    for rep in range(2):
        for ovl in (0, 1):
            with pkg.Engine(nx, ny) as e:
                e.set_option("window_overlap", ovl)
                e.set_mask(mask); e.init_equilibrium(0.06); e.step(600, 0.58, 0.06); e.sync()
                n = 1200 if nx * ny < 3e6 else 408
                us = min(e.step_timed(n, 0.58, 0.06) for _ in range(3)) / n * 1e3
                res.setdefault(ovl, []).append(us)
                info = (int(e.get_option("fuse_active")), int(e.get_option("fuse_depth")), int(e.get_option("fuse_units")), int(e.get_option("window_overlap")))
    print(f"{nx}x{ny}: tiling {min(res[0]):.2f} us/step, overlapping {min(res[1]):.2f} us/step  ({nx * ny / min(res[0]) / 1e3:.1f} vs {nx * ny / min(res[1]) / 1e3:.1f} GLUPS)  plan (active, depth, units, overlap) of the last run {info}", flush=True)
