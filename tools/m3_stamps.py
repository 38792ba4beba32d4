"""Developer experiment: per-segment clocks of the lean three-step loop (library built with -DWT_M3_STAMPS into tools/ab/lib_stamps.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg
lib = pkg.load_library()
nx = ny = 4096
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(30, 0.58, 0.06); e.sync()
    out = (ctypes.c_ulonglong * 8)()
    lib.wt_debug_m3_stamps(out, 1)
    ms = e.step_timed(300, 0.58, 0.06)
    lib.wt_debug_m3_stamps(out, 0)
    v = [int(x) for x in out]
    iters, units = v[6], v[7]
    names = ["issue prefetch + halo loads + LDS fetch", "shift + STEP1 (level 1)", "stage 1 (level 2)", "stage 2 (level 3)", "wait for prefetched column", "issue stores + seam flush"]
    tot = sum(v[:6])
    print(f"300 steps in {ms:.2f} ms ({ms/300*1000:.1f} us/step); lean units {units}, iterations {iters}; clocks per iteration {tot/iters:.0f}")
    for n, c in zip(names, v[:6]):
        print(f"  {n:42s} {c/iters:8.0f} clocks/iteration  {100*c/tot:5.1f} %")
