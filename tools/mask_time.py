import sys, time
sys.path.insert(0, '.')
import numpy as np
import airfoil_cfd_tool_amd as pkg
for nx, ny in ((1024, 512), (4096, 4096)):
    masks = [pkg.geometry.build_geometry(nx, ny, a, None, "naca2412").mask for a in (4.0, 4.5, 5.0, 5.5, 6.0)]
    with pkg.Engine(nx, ny) as e:
        e.set_mask(masks[0]); e.init_equilibrium(0.06); e.step(10, 0.58, 0.06); e.sync()
        ts = []
        for k in range(20):
            t0 = time.perf_counter(); e.set_mask(masks[k % 5]); ts.append(time.perf_counter() - t0)
        print(nx, ny, "set_mask ms: min %.3f median %.3f max %.3f" % (min(ts)*1e3, sorted(ts)[10]*1e3, max(ts)*1e3), "fuse_active", e.get_option("fuse_active"))
        t0 = time.perf_counter(); g = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412"); print("  build_geometry (host rasterMask) ms", (time.perf_counter()-t0)*1e3)
