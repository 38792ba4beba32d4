"""Runs ON THE GPU BOX: trace of the measured refinement (WT_TUNE_TRACE=1) on the whole lattice and on slab 3 of the 8-way split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
def run(sub, tag, dtype="float32"):
    with pkg.Engine(sub.shape[1], ny, dtype=dtype) as e:
        e.set_mask(sub); e.init_equilibrium(0.06); e.step(24, 0.58, 0.06)
        us = min(e.step_timed(408, 0.58, 0.06) for _ in range(3)) / 408 * 1e3
        print(f"{tag}: {us:.2f} us/step, gain {e.get_option('tune_gain'):.3f} after {int(e.get_option('tune_rounds'))} rounds", flush=True)
run(mask, "whole")
for r in (2, 3, 4):
    lo, hi = r * 512 - 16, (r + 1) * 512 + 16
    run(np.ascontiguousarray(mask[:, lo:hi]), f"slab {r}/8")
run(np.ascontiguousarray(mask[:, 1024 - 16:2048 + 16]), "slab 1/4")
run(mask, "whole fp64", "float64")
