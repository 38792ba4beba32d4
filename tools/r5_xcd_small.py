"""Developer experiment: the XCD-contiguous workgroup order on small whole lattices with overlapping windows (WT_XCD_ORDER, experiment build).
    WT_XCD_ORDER=0|1 python3 tools/r5_xcd_small.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", "lib_knobs.so")
capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
for nx, ny in ((1024, 512), (2048, 1024), (2048, 2048), (3584, 2048)):
    mask = pkg.geometry.build_geometry(nx, ny, 5.0, None, "naca2412").mask
    best = 1e9
    for rep in range(2):
        with pkg.Engine(nx, ny) as e:
            e.set_mask(mask); e.init_equilibrium(0.06); e.step(600, 0.58, 0.06); e.sync()
            n = 1200 if nx * ny < 3e6 else 408
            best = min(best, min(e.step_timed(n, 0.58, 0.06) for _ in range(3)) / n * 1e3)
            ov = int(e.get_option("window_overlap"))
    print(f"WT_XCD_ORDER={os.environ.get('WT_XCD_ORDER', '-')} {nx}x{ny} overlap {ov}: {best:.2f} us/step", flush=True)
