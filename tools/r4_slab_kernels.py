"""Runs ON THE GPU BOX under `rocprofv3 --kernel-trace`: slab `rank` of the equal 8-way split of the bench tunnel as a stand-alone handle (owned + 16 ghost
columns a side, planned like the split's narrowest slab), 400 steps — the per-pass kernels of a BODY-holding slab.  argv: ranks (default 0 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
edges = pkg.slab_edges(nx, 8)
for r in [int(a) for a in sys.argv[1:]] or [0, 3]:
    lo, hi = max(0, edges[r] - 16), min(nx, edges[r + 1] + 16)
    with pkg.Engine(hi - lo, ny) as e:
        e.set_option("plan_columns", 512 + 16)
        e.set_mask(np.ascontiguousarray(mask[:, lo:hi])); e.init_equilibrium(0.06); e.step(200, 0.58, 0.06)
        ms = e.step_timed(400, 0.58, 0.06) / 400
        print(f"slab {r} ({hi - lo} columns): {ms * 1e3:.2f} us per step, units {int(e.get_option('fuse_units'))} ({int(e.get_option('chain_units'))} chain), general tiles {int(e.get_option('fuse_tiles_general'))}", flush=True)
