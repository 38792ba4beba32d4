"""Developer experiment: XCD-aware workgroup order (WT_XCD_ORDER, experiment build tools/ab/lib_knobs.so) on both window layouts.
    python3 tools/r5_xcd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
order = os.environ.get("WT_XCD_ORDER", "-")
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", "lib_knobs.so")
capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
full = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
tests = [("whole 4096^2", full, {}), ("slab 0/8 (528 columns)", np.ascontiguousarray(full[:, 0:528]), {"plan_columns": 541}),
         ("slab 3/8 (544 columns)", np.ascontiguousarray(full[:, 1520:2064]), {"plan_columns": 541})]
for name, mask, extra in tests:
    for ovl in (0, 1):
        with pkg.Engine(mask.shape[1], mask.shape[0]) as e:
            for k, v in extra.items():
                e.set_option(k, v)
            e.set_option("window_overlap", ovl)
            e.set_mask(mask); e.init_equilibrium(0.06); e.step(400, 0.58, 0.06); e.sync()
            us = min(e.step_timed(408, 0.58, 0.06) for _ in range(2)) / 408 * 1e3
            print(f"WT_XCD_ORDER={order} {name}: overlap {int(e.get_option('window_overlap'))}, {int(e.get_option('fuse_units'))} units: {us:.2f} us per step", flush=True)
