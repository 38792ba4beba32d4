"""Developer experiment / check: overlapping windows (option window_overlap = 1) against windows that tile the column (0) and against single steps:
bit-identical populations and macro fields, and what a pass costs either way.
    python3 tools/r5_overlap.py [time]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg


def run(nx, ny, mask, nsteps, opts, dtype="float32", tau=0.58):
    with pkg.Engine(nx, ny, dtype=dtype) as e:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_mask(mask); e.init_equilibrium(0.06)
        e.step(nsteps, tau, 0.06)
        info = {k: e.get_option(k) for k in ("window_overlap", "fuse_depth", "fuse_units", "fuse_active", "single_steps", "passes")}
        return e.read_f(), e.read_macro(), info


bad = 0
cases = [(320, 256, "naca0012", 4.0, 37), (640, 512, "naca2412", 5.0, 41), (768, 1000, "naca4412", 12.0, 29), (512, 120, "naca0012", 0.0, 23),
         (544, 1366, "naca6409", 10.0, 31), (1024, 2048, "naca6409", 10.0, 45)]
for nx, ny, shape, aoa, nsteps in cases:
    mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
    ref_f, ref_m, _ = run(nx, ny, mask, nsteps, {"fuse_steps": 0})
    for depth in (4, 3):
        for ovl in (0, 1):
            f, m, info = run(nx, ny, mask, nsteps, {"fuse_steps": 2, "fuse_depth": depth, "window_overlap": ovl})
            same = np.array_equal(f.view(np.uint32), ref_f.view(np.uint32)) and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(m, ref_m))
            bad += not same
            print(f"{nx}x{ny} {shape} {nsteps} steps, depth {depth}, overlap asked {ovl} -> {info}: {'identical' if same else 'DIFFERENT'}", flush=True)
            if not same:
                d = np.argwhere(f.view(np.uint32) != ref_f.view(np.uint32))
                print("   first differences (k, j, i):", d[:6].tolist(), " count", len(d), " rows", sorted(set(d[:, 1].tolist()))[:20])
print("MISMATCHES:", bad)
if len(sys.argv) > 1:
    full = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
    tests = [("whole 4096^2", full, {}), ("slab 0/8 (528 columns)", np.ascontiguousarray(full[:, 0:528]), {"plan_columns": 541}),
             ("slab 3/8 (544 columns)", np.ascontiguousarray(full[:, 1520:2064]), {"plan_columns": 541})]
    for name, mask, extra in tests:
        for rep in range(2):
            for ovl in (0, 1):
                with pkg.Engine(mask.shape[1], mask.shape[0]) as e:
                    for k, v in extra.items():
                        e.set_option(k, v)
                    e.set_option("window_overlap", ovl)
                    e.set_mask(mask); e.init_equilibrium(0.06); e.step(400, 0.58, 0.06); e.sync()
                    us = e.step_timed(408, 0.58, 0.06) / 408 * 1e3
                    print(f"{name}: overlap {int(e.get_option('window_overlap'))}, {int(e.get_option('fuse_units'))} units, depth {int(e.get_option('fuse_depth'))}: {us:.2f} us per step", flush=True)
sys.exit(1 if bad else 0)
