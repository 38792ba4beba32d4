"""Developer experiment: how long does every marching unit of one pass take?  Library built with -DWT_UNIT_CLOCKS into
tools/ab/lib_clocks.so (tools/build_ab.sh); per-unit s_memtime stamps of the LAST pass, summarised by unit kind.
    python3 tools/unit_clocks.py NX NY DEPTH CHAIN [body]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", os.environ.get("WT_AB_LIB", "lib_clocks.so"))
lib = capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
nx, ny, depth, chain = (int(v) for v in sys.argv[1:5])
body = len(sys.argv) > 5
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask if body else np.zeros((ny, nx), np.uint8)
if os.environ.get("WT_SLAB"):        # WT_SLAB="r P": the columns slab r of a P-way split of this lattice holds (halo 16), as a stand-alone lattice
    r_, P_ = (int(v) for v in os.environ["WT_SLAB"].split())
    lo_, hi_ = max(0, r_ * nx // P_ - 16), min(nx, (r_ + 1) * nx // P_ + 16)
    mask = np.ascontiguousarray(mask[:, lo_:hi_]); nx = hi_ - lo_
with pkg.Engine(nx, ny) as e:
    e.set_option("chain", chain); e.set_option("fuse_depth", depth)
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(10 * depth, 0.58, 0.06); e.sync()
    ms = e.step_timed(40 * depth, 0.58, 0.06)
    cap = 16384
    clk = (ctypes.c_ulonglong * (2 * cap))(); units = (ctypes.c_int * (4 * cap))()
    lib.wt_debug_unit_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    n = lib.wt_debug_unit_clocks(e._h, clk, units, cap)
    c = np.array(clk[:2 * n], dtype=np.uint64).reshape(n, 2).astype(np.int64)
    u = np.array(units[:4 * n]).reshape(n, 4)
    t0 = c[:, 0].min()
    L = u[:, 1] - u[:, 0]
    live = L > 0
    dur = (c[:, 1] - c[:, 0]) / 100.0          # s_memtime ticks at 100 MHz -> us
    end = (c[:, 1] - t0) / 100.0
    start = (c[:, 0] - t0) / 100.0
    print(f"{nx}x{ny} depth {depth} chain {chain} body {int(body)}: {ms / 40 * 1000:.1f} us per pass; {n} units ({int(live.sum())} live); "
          f"last unit ends at {end[live].max():.1f} us, median start {np.median(start[live]):.1f} us")
    kinds = {"chain inner": (u[:, 3] & 2 != 0) & (u[:, 3] & 8 != 0), "chain outer": (u[:, 3] & 2 != 0) & (u[:, 3] & 8 == 0),
             "solo": (u[:, 3] & 2 == 0) & live}
    for name, sel in kinds.items():
        if sel.any():
            d = dur[sel]
            print(f"  {name:12s} {int(sel.sum()):5d} units, columns {L[sel].min()}..{L[sel].max()} (mean {L[sel].mean():.1f}); duration us: "
                  f"min {d.min():.1f} median {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} max {d.max():.1f}; ends: median {np.median(end[sel]):.1f} max {end[sel].max():.1f}")
    solo = kinds["solo"]
    if solo.any():
        for Lv in sorted(set(L[solo])):
            s2 = solo & (L == Lv)
            print(f"    solo with {Lv:2d} columns: {int(s2.sum()):4d} units, median {np.median(dur[s2]):.1f} max {dur[s2].max():.1f} us")
    # where are the slow units?  by XCD (workgroup index mod 8), by window, by column range
    blk = np.arange(n) // 4
    sel = live
    print("  by XCD (workgroup % 8): median / max duration")
    print("    " + "  ".join(f"{int(np.median(dur[sel & (blk % 8 == x)]))}/{int(dur[sel & (blk % 8 == x)].max())}" for x in range(8)))
    ws = sorted(set(u[sel, 2]))
    print("  by window: median / max")
    print("    " + "  ".join(f"w{w}:{int(np.median(dur[sel & (u[:, 2] == w)]))}/{int(dur[sel & (u[:, 2] == w)].max())}" for w in ws))
    qs = np.quantile(u[sel, 0], [0, 0.25, 0.5, 0.75, 1.0])
    print("  by first column (quartiles): median / max")
    for a, b in zip(qs[:-1], qs[1:]):
        s3 = sel & (u[:, 0] >= a) & (u[:, 0] <= b)
        print(f"    ia {int(a):5d}..{int(b):5d}: {int(np.median(dur[s3]))}/{int(dur[s3].max())}")
    st = (c[:, 0] - c[:, 0].min())
    print("  start stamps relative to the earliest (same-XCD counters only are comparable), per XCD: median / max start offset in ticks/100")
    for x in range(8):
        s4 = sel & (blk % 8 == x)
        s0 = c[s4, 0] - c[s4, 0].min()
        e0 = c[s4, 1] - c[s4, 0].min()
        print(f"    XCD {x}: starts median {np.median(s0)/100:.0f} max {s0.max()/100:.0f}; ends median {np.median(e0)/100:.0f} max {e0.max()/100:.0f}")
