"""Developer experiment: steps per pass (3 / 4) on overlapping windows, stand-alone slabs of the 8-way split of 4096^2 planned like the split.
    python3 tools/r5_ovl_depth.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg
full = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
tests = [("slab 0/8 (528 columns)", np.ascontiguousarray(full[:, 0:528])), ("slab 3/8 (544 columns)", np.ascontiguousarray(full[:, 1520:2064])),
         ("slab of a 16-way split (288 columns)", np.ascontiguousarray(full[:, 1776:2064]))]
for name, mask in tests:
    for rep in range(2):
        for depth in (4, 3):
            for ovl in (1, 0):
                with pkg.Engine(mask.shape[1], mask.shape[0]) as e:
                    e.set_option("plan_columns", min(541, mask.shape[1])); e.set_option("fuse_depth", depth); e.set_option("window_overlap", ovl)
                    e.set_mask(mask); e.init_equilibrium(0.06); e.step(400, 0.58, 0.06); e.sync()
                    us = min(e.step_timed(408, 0.58, 0.06) for _ in range(2)) / 408 * 1e3
                    print(f"{name}: depth {int(e.get_option('fuse_depth'))} overlap {int(e.get_option('window_overlap'))}, {int(e.get_option('fuse_units'))} units: {us:.2f} us per step", flush=True)
