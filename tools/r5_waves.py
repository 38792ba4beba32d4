"""Developer experiment: one marching wave per SIMD against two (WT_MARCH_WAVES, experiment build tools/ab/lib_knobs.so): what does a lone wave reach?
    python3 tools/r5_waves.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", "lib_knobs.so")
capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
full = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
cases = [("whole 4096^2", full), ("slab 0/8 (528 columns)", np.ascontiguousarray(full[:, 0:528])), ("slab 3/8 (544 columns)", np.ascontiguousarray(full[:, 1520:2064]))]
for name, mask in cases:
    for rep in range(2):
        for waves in (2, 1):
            os.environ["WT_MARCH_WAVES"] = str(waves)
            with pkg.Engine(mask.shape[1], mask.shape[0]) as e:
                e.set_mask(mask); e.init_equilibrium(0.06); e.step(400, 0.58, 0.06); e.sync()
                us = e.step_timed(408, 0.58, 0.06) / 408 * 1e3
                print(f"{name}: {waves} wave(s) per SIMD: {int(e.get_option('fuse_units'))} units, depth {int(e.get_option('fuse_depth'))}: {us:.2f} us per step", flush=True)
