"""Runs ON THE GPU BOX: a long run of a locally linked slab group (overlapping windows, XCD order, prebuilt trimmed lists, hosts' default halo) against the
single lattice stepped by k_step: the same bits after thousands of steps, through every kind of remainder, for each refresh mode.
    python tools/r5_soak_slabs.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg
from airfoil_cfd_tool_amd.distributed import DEFAULT_HALO

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
nx, ny, shape, aoa, tau, P = 3072, 2048, "naca6409", 10.0, 0.56, 4
mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
chunks = (steps // 3, steps // 3 + 1, steps - 2 * (steps // 3) - 1)
with pkg.Engine(nx, ny) as e:
    e.set_option("fuse_steps", 0)
    e.set_mask(mask); e.init_equilibrium(0.08)
    for c in chunks:
        e.step(c, tau, 0.08)
    ref_f = e.read_f(); ref_m = e.read_macro()
for refresh in (0, 2, 1):
    es = [pkg.Engine(nx, ny, rank=r, nranks=P, halo=DEFAULT_HALO) for r in range(P)]
    try:
        pkg.Engine.link_local(es)
        for s in es:
            s.set_option("refresh", refresh)
            s.set_mask(mask); s.init_equilibrium(0.08)
        t0 = time.time()
        for c in chunks:
            pkg.Engine.step_group(es, c, tau, 0.08)
        f = np.concatenate([s.read_f() for s in es], axis=2)
        m = [np.concatenate([s.read_macro()[k] for s in es], axis=1) for k in range(3)]
        same = np.array_equal(f.view(np.uint32), ref_f.view(np.uint32)) and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(m, ref_m))
        print(f"{shape} {nx}x{ny} fp32 tau {tau}, {P} local slabs, halo {DEFAULT_HALO}, refresh {refresh}, window_overlap {[int(s.get_option('window_overlap')) for s in es]}: "
              f"{steps} steps ({int(es[1].get_option('passes'))} passes, {int(es[1].get_option('single_steps'))} single steps, {int(es[1].get_option('trimmed_passes'))} trimmed, "
              f"{time.time() - t0:.1f} s) vs k_step on the single lattice: {'BIT-IDENTICAL' if same else 'DIFFER'}; finite {bool(np.isfinite(f).all())}", flush=True)
    finally:
        for s in es:
            s.close()
