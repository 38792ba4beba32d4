"""Runs ON THE GPU BOX: does a slab of a locally linked group cost what its stand-alone stand-in (distributed.measure_slab_cost) says?  All slabs on one GPU, so
the group's wall time per step should be about the SUM of the stand-alone costs.  Equal widths against two uneven splits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096; halo = 16; P = 8
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
splits = {"equal": pkg.slab_edges(nx, P),
          "cut by cost": [0, 682, 1206, 1634, 2033, 2452, 2884, 3418, 4096],
          "mild": [0, 560, 1080, 1580, 2060, 2540, 3030, 3540, 4096]}
only = os.environ.get("WT_SPLIT")
for name, edges in splits.items():
    if only and name != only:
        continue
    alone = [pkg.measure_slab_cost(mask, edges, r, halo, steps=408) for r in range(P)] if not only else [0.0] * P
    es = [pkg.Engine(nx, ny, rank=r, nranks=P, halo=halo, edges=edges) for r in range(P)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.init_equilibrium(0.06)
        pkg.Engine.step_group(es, 34, 0.58, 0.06)
        for e in es: e.sync()
        t0 = time.perf_counter()
        pkg.Engine.step_group(es, 408, 0.58, 0.06)
        for e in es: e.sync()
        wall = (time.perf_counter() - t0) / 408 * 1e6
        info = [(int(e.get_option("fuse_depth")), int(e.get_option("fuse_units")), int(e.get_option("chain_units")), round(e.get_option("tune_gain"), 3)) for e in es]
    finally:
        for e in es: e.close()
    print(f"{name}: widths {[b - a for a, b in zip(edges[:-1], edges[1:])]}\n   alone {[round(a, 2) for a in alone]} sum {sum(alone):.1f} | group wall {wall:.1f} us/step"
          f" | (depth, units, chain units, tune gain) {info}", flush=True)
