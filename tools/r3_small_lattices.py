"""Runs ON THE GPU BOX: small lattices (BASELINE configs 0-1 and neighbours): single steps against the automatic plan and forced three / four steps per pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
for (nx, ny, shape, aoa) in ((256, 128, "naca0012", 0.0), (512, 256, "naca2412", 5.0), (1024, 512, "naca2412", 5.0), (2048, 1024, "naca2412", 5.0), (1024, 1024, "naca2412", 5.0), (600, 400, "naca4412", 6.0)):
    mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
    row = []
    for label, opts in (("single", {"fuse_steps": 0}), ("auto", {}), ("three", {"fuse_depth": 3, "fuse_steps": 2}), ("four", {"fuse_depth": 4, "fuse_steps": 2})):
        with pkg.Engine(nx, ny) as e:
            for k, v in opts.items():
                e.set_option(k, v)
            e.set_mask(mask); e.init_equilibrium(0.06); e.step(48, 0.58, 0.06)
            us = min(e.step_timed(1200, 0.58, 0.06) for _ in range(2)) / 1200 * 1e3
            d = int(e.get_option("fuse_depth")) if e.get_option("fuse_active") else 0
            row.append(f"{label} (depth {d}, {int(e.get_option('fuse_units')) if d else 0} units) {us:.2f} us = {nx * ny / us / 1e3:.1f} GLUPS")
    print(f"{nx}x{ny}: " + " | ".join(row), flush=True)
