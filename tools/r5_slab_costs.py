"""Runs ON THE GPU BOX: what does each slab of the REAL 8-way split cost per step, with the refresh machinery running and the exchange priced?

    python tools/r5_slab_costs.py CONFIG [N] [halo]      CONFIG 2 = 4096x4096 NACA 6409 10 deg (the metric's lattice), 3 = 16384x4096 NACA 0012 8 deg

Round 4's projection (tools/r4_slab_costs.py) stepped a stand-alone lattice of the slab's width and ASSUMED the exchange hides.  Here every slab of
the split is a real slab handle, alone on the GPU and linked to itself (distributed.measure_slab_real: ghost refreshes, trimmed ghost marching, the
refresh mode under test — all of it runs, with a copy kernel as the exchange), and the exchange over xGMI is ADDED from bench.py's stated model
(exchange_model_us = 12 us + 9 x halo x pitch x element size / 76.8 GB/s per side, sides in parallel): per refresh, what of the model does not fit
into the compute that runs beside the exchange (`interior_us`, measured) is exposed.  Split: equal widths, then cut by the measured costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg
import bench

CFG = {2: (4096, 4096, "naca6409", 10.0), 3: (16384, 4096, "naca0012", 8.0)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
halo = int(sys.argv[3]) if len(sys.argv) > 3 else 29
nx, ny, shape, aoa = CFG[cfg]
modes = [int(m) for m in os.environ.get("WT_REFRESH_MODES", "0 2").split()]
rounds = int(os.environ.get("WT_BALANCE_ROUNDS", "3"))
mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(300, 0.58, 0.06)
    whole = e.step_timed(408, 0.58, 0.06) / 408 * 1e3
print(f"BASELINE configs[{cfg}] {shape} {nx}x{ny}: whole lattice on one GPU {whole:.2f} us/step = {nx * ny / whole / 1e3:.1f} GLUPS; halo {halo}", flush=True)


def priced(m, mode):
    """measured us per step + the exposed part of the modelled exchange per refresh, spread over the steps between two refreshes"""
    mod = bench.exchange_model(halo, ny, 4, m["sides"])
    cycle = m["steps"] / max(1, m["exchanges"])
    exposed = max(0.0, mod["exchange_model_us"] - m["interior_us"]) if m["sides"] else 0.0
    # the copy kernel that stands in for the exchange is part of the measurement already; what of IT was exposed is replaced, not added twice
    return m["us_per_step"] + (exposed - min(exposed, m["exposed_us"])) / cycle, exposed, mod["exchange_model_us"], cycle


for mode in modes:
    opts = {"refresh": mode}
    if os.environ.get("WT_SLAB_TRIM"):                         # experiments: trimmed ghost marching on / off
        opts["trim_ghosts"] = int(os.environ["WT_SLAB_TRIM"])
    if os.environ.get("WT_SLAB_OVERLAP"):                      # experiments: force the window layout of the slabs (0 tiling, 1 overlapping)
        opts["window_overlap"] = int(os.environ["WT_SLAB_OVERLAP"])
    measure = lambda ed: [pkg.measure_slab_real(mask, ed, r, halo, options=opts) for r in range(P)]
    hist = []
    best, _ = pkg.balance_split(nx, P, max(2 * halo, 64), lambda ed: (hist.append((list(ed), measure(ed))) or [priced(m, mode)[0] for m in hist[-1][1]]), rounds)
    for k, (ed, ms) in enumerate(hist):
        pr = [priced(m, mode) for m in ms]
        worst_raw, worst = max(m["us_per_step"] for m in ms), max(p[0] for p in pr)
        tag = "equal widths" if k == 0 else f"cut by cost, round {k}"
        print(f"refresh {mode}, N = {P}, {tag}: widths {[b - a for a, b in zip(ed[:-1], ed[1:])]}" + ("   <- kept" if ed == best else ""))
        print("    measured us/step (copy-kernel exchange)   " + "  ".join(f"{m['us_per_step']:6.2f}" for m in ms) + f"   slowest {worst_raw:.2f} -> {whole / worst_raw:.2f} x")
        print("    compute beside the exchange, us           " + "  ".join(f"{m['interior_us']:6.1f}" for m in ms))
        print(f"    modelled exchange {pr[1][2]:.1f} us per refresh of {pr[1][3]:.0f} steps; exposed per refresh, us      " + "  ".join(f"{p[1]:6.1f}" for p in pr))
        print("    us/step with the modelled exchange        " + "  ".join(f"{p[0]:6.2f}" for p in pr) +
              f"   slowest {worst:.2f} -> {nx * ny / worst / 1e3:.0f} GLUPS = {whole / worst:.2f} x the one-GPU run", flush=True)
    m1 = hist[-1][1][1]
    print(f"    (slab 1: {m1['passes']} passes, {m1['single_steps']} single steps, {m1['fused_renewals']} fused renewals, {m1['exchanges']} exchanges in {m1['steps']} steps)", flush=True)
