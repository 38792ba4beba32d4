"""Runs ON THE GPU BOX: what does trimmed ghost marching (option trim_ghosts) buy a locally linked 8-slab group of the bench tunnel?  All slabs share
one GPU, so the group's wall time per step is the SUM of the slabs' work (plus the refresh steps): the saved ghost columns show up directly.
usage: python tools/r4_trim_ab.py [halo ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096
P = 8
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
for halo in [int(a) for a in sys.argv[1:]] or [17, 25, 33]:
    for rep in range(2):
        for trim, refresh in ((0, 0), (1, 0), (1, 1)) if os.environ.get("WT_REFRESH_AB") else ((0, 0), (1, 0)):
            es = [pkg.Engine(nx, ny, rank=r, nranks=P, halo=halo) for r in range(P)]
            try:
                pkg.Engine.link_local(es)
                for e in es:
                    e.set_option("trim_ghosts", trim)
                    e.set_option("refresh", refresh)
                    e.set_mask(mask); e.init_equilibrium(0.06)
                pkg.Engine.step_group(es, 10 * halo, 0.58, 0.06)
                for e in es:
                    e.sync()
                n = 40 * halo
                t0 = time.perf_counter()
                pkg.Engine.step_group(es, n, 0.58, 0.06)
                for e in es:
                    e.sync()
                us = (time.perf_counter() - t0) / n * 1e6
                print(f"halo {halo:2d} trim {trim} refresh {refresh}: {us:7.2f} us per step of the group ({nx * ny / us / 1e3:.0f} GLUPS on one GPU), trimmed passes {int(es[1].get_option('trimmed_passes'))} of {int(es[1].get_option('passes'))}, single steps {int(es[1].get_option('single_steps'))}, boundary exchanges {int(es[1].get_option('boundary_exchanges'))}", flush=True)
            finally:
                for e in es:
                    e.close()
