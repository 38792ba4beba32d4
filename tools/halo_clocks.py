"""Developer experiment: where does k_halo4 spend its clocks?  (library built with -DWT_UNIT_CLOCKS: tools/build_ab.sh)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", "lib_clocks.so")
lib = capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
nx, ny = int(sys.argv[1]), int(sys.argv[2])
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(40, 0.58, 0.06); e.sync()
    out = (ctypes.c_ulonglong * 8)()
    lib.wt_debug_halo_clocks(out, 1)
    e.step(160, 0.58, 0.06); e.sync()
    lib.wt_debug_halo_clocks(out, 0)
    v = [int(x) for x in out]; n = max(1, v[7])
    names = ["level 1 + level-0 words (loads, collision, LDS)", "barrier 1", "level 2", "barrier 2", "level 3 + the lines' words in LDS", "barrier 3 + the lines' store"]
    print(f"{nx}x{ny}: {n} workgroups of k_halo4 in 40 passes; clocks of thread 0 per workgroup:")
    for nm, c in zip(names, v[:6]):
        print(f"  {nm:36s} {c / n:9.0f}")
    print(f"  total {sum(v[:6]) / n:.0f} clocks = {sum(v[:6]) / n / 2.4e3:.2f} us at 2.4 GHz")
