"""Developer experiment: longer units for the workgroups dispatched first (WT_MARCH_REV=0 WT_ASYM=gamma, experiment build tools/ab/lib_knobs.so, WT_TUNE=0).
    WT_MARCH_REV=0 WT_ASYM=0.12 WT_TUNE=0 python3 tools/r5_asym.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", "lib_knobs.so")
capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
tag = f"REV={os.environ.get('WT_MARCH_REV', '-')} ASYM={os.environ.get('WT_ASYM', '-')} TUNE={os.environ.get('WT_TUNE', '-')}"
full = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
tests = [("whole 4096^2", full, {}), ("slab 0/8 (528 columns)", np.ascontiguousarray(full[:, 0:528]), {"plan_columns": 541})]
for name, mask, extra in tests:
    with pkg.Engine(mask.shape[1], mask.shape[0]) as e:
        for k, v in extra.items():
            e.set_option(k, v)
        e.set_mask(mask); e.init_equilibrium(0.06); e.step(400, 0.58, 0.06); e.sync()
        us = min(e.step_timed(408, 0.58, 0.06) for _ in range(3)) / 408 * 1e3
        print(f"{tag} {name}: {int(e.get_option('fuse_units'))} units: {us:.2f} us per step", flush=True)
