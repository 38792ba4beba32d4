"""Runs ON THE GPU BOX: what would each slab of an N-way split of the bench lattice cost per step?  For every rank r a stand-alone handle of the
slab's local width (owned + ghost columns) is stepped on the mask columns that slab holds (so the body pieces are the real ones; the handle's
own edges are treated as inlet / outlet, which a slab's are not — a small pessimism), 408 timed steps each
(airfoil_cfd_tool_amd.distributed.measure_slab_cost).  First equal widths, then the slabs cut by measured cost (balance_split, what
`bench.py --gpus N` does before a strong-scaling run).  The slowest slab sets the pace of the N-GPU run; this is still a one-GPU PROJECTION
(no exchange, no refresh steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg

nx = ny = 4096
halo = 16
rounds = int(os.environ.get("WT_BALANCE_ROUNDS", "4"))
opts = {"fuse_depth": int(os.environ["WT_SLAB_DEPTH"])} if os.environ.get("WT_SLAB_DEPTH") else None
splits = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(24, 0.58, 0.06)
    whole = e.step_timed(408, 0.58, 0.06) / 408 * 1e3
    gain = e.get_option("tune_gain")
print(f"whole 4096^2 lattice on one GPU: {whole:.2f} us/step = {nx * ny / whole / 1e3:.1f} GLUPS (tune gain {gain:.3f})")
for P in splits:
    best, hist = pkg.balance_split(nx, P, 32, lambda ed: [pkg.measure_slab_cost(mask, ed, r, halo, steps=408, options=opts) for r in range(P)], rounds)
    for k, (ed, cost) in enumerate(hist):
        worst = max(cost)
        tag = "equal widths" if k == 0 else f"cut by cost, round {k}"
        print(f"N = {P} {tag}: widths {[b - a for a, b in zip(ed[:-1], ed[1:])]}")
        print(f"        per-slab us/step " + "  ".join(f"{c:.2f}" for c in cost) +
              f"   slowest {worst:.2f} -> {nx * ny / worst / 1e3:.0f} GLUPS if the exchange hides = {whole / worst:.2f} x the one-GPU run"
              + ("   <- kept" if ed == best and k > 0 else ""))
