"""Developer experiment: the size threshold of overlapping windows on whole lattices, and steps per pass on that layout for small ones.
    python3 tools/r5_ovl_sizes2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg
def run(nx, ny, mask, opts):
    best = 1e9
    for rep in range(2):
        with pkg.Engine(nx, ny) as e:
            for k, v in opts.items():
                e.set_option(k, v)
            e.set_mask(mask); e.init_equilibrium(0.06); e.step(600, 0.58, 0.06); e.sync()
            n = 1200 if nx * ny < 3e6 else 408
            best = min(best, min(e.step_timed(n, 0.58, 0.06) for _ in range(3)) / n * 1e3)
            info = (int(e.get_option("fuse_active")), int(e.get_option("fuse_depth")), int(e.get_option("fuse_units")), int(e.get_option("window_overlap")))
    return best, info
for nx, ny in ((2560, 2048), (3072, 2048), (2048, 3072), (3584, 2048), (768, 384), (1536, 768)):
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask
    a, ia = run(nx, ny, mask, {"window_overlap": 0}); b, ib = run(nx, ny, mask, {"window_overlap": 1})
    print(f"{nx}x{ny} ({nx * ny / 1e6:.1f} M sites): tiling {a:.2f} us/step {ia}, overlapping {b:.2f} us/step {ib}", flush=True)
for nx, ny in ((1024, 512), (2048, 1024), (1536, 768), (768, 384)):
    mask = pkg.geometry.build_geometry(nx, ny, 5.0, None, "naca2412").mask
    for depth in (3, 4):
        b, ib = run(nx, ny, mask, {"window_overlap": 1, "fuse_depth": depth})
        print(f"{nx}x{ny} overlapping, fuse_depth {depth}: {b:.2f} us/step {ib}", flush=True)
