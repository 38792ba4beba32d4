"""Runs ON THE GPU BOX: what would each slab of an N-way split cost per step?  tools/r3_slab_costs.py for any BASELINE configuration:

    python tools/r4_slab_costs.py CONFIG N [N ...]        CONFIG 2 = 4096x4096 NACA 6409 10 deg (strong scaling of the metric's lattice),
                                                          CONFIG 3 = 16384x4096 NACA 0012 8 deg (BASELINE configs[3], the 8-GPU case)

For every rank a stand-alone handle of the slab's local width (owned + ghost columns) is stepped on the mask columns that slab holds, planned like
the narrowest slab of the split (airfoil_cfd_tool_amd.distributed.measure_slab_cost); equal widths first, then the slabs cut by measured cost
(balance_split, what `bench.py --gpus N` does before a strong-scaling run).  A one-GPU PROJECTION: no exchange, no refresh steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg

CFG = {2: (4096, 4096, "naca6409", 10.0), 3: (16384, 4096, "naca0012", 8.0)}
cfg = int(sys.argv[1])
nx, ny, shape, aoa = CFG[cfg]
halo = 16
rounds = int(os.environ.get("WT_BALANCE_ROUNDS", "4"))
splits = [int(a) for a in sys.argv[2:]] or [8]
mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(200, 0.58, 0.06)
    whole = e.step_timed(408, 0.58, 0.06) / 408 * 1e3
print(f"BASELINE configs[{cfg}] {shape} {nx}x{ny}: whole lattice on one GPU {whole:.2f} us/step = {nx * ny / whole / 1e3:.1f} GLUPS", flush=True)
for P in splits:
    best, hist = pkg.balance_split(nx, P, 32, lambda ed: [pkg.measure_slab_cost(mask, ed, r, halo, steps=408) for r in range(P)], rounds)
    for k, (ed, cost) in enumerate(hist):
        worst = max(cost)
        tag = "equal widths" if k == 0 else f"cut by cost, round {k}"
        print(f"N = {P} {tag}: widths {[b - a for a, b in zip(ed[:-1], ed[1:])]}")
        print("        per-slab us/step " + "  ".join(f"{c:.2f}" for c in cost) +
              f"   slowest {worst:.2f} -> {nx * ny / worst / 1e3:.0f} GLUPS if the exchange hides = {whole / worst:.2f} x the one-GPU run of the same lattice"
              + ("   <- kept" if ed == best else ""), flush=True)
