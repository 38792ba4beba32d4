"""Developer experiment: WHEN, inside one launch of the marching kernel, does every unit start and end?  Library built with
-DWT_UNIT_CLOCKS -DWT_CLOCK_REALTIME into tools/ab/lib_timeline.so (s_memrealtime: the constant 100 MHz counter all XCDs share); stamps of the LAST pass.
    python3 tools/unit_timeline.py NX NY [body]        WT_SLAB="r P": the columns of slab r of a P-way split (+ 16 ghost columns), as a stand-alone lattice"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", "lib_timeline.so")
lib = capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
nx, ny = int(sys.argv[1]), int(sys.argv[2])
body = len(sys.argv) > 3
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask if body else np.zeros((ny, nx), np.uint8)
if os.environ.get("WT_SLAB"):
    r_, P_ = (int(v) for v in os.environ["WT_SLAB"].split())
    lo_, hi_ = max(0, r_ * nx // P_ - 16), min(nx, (r_ + 1) * nx // P_ + 16)
    mask = np.ascontiguousarray(mask[:, lo_:hi_]); nx = hi_ - lo_
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(400, 0.58, 0.06); e.sync()
    ms = e.step_timed(400, 0.58, 0.06)
    cap = 16384
    clk = (ctypes.c_ulonglong * (2 * cap))(); units = (ctypes.c_int * (4 * cap))()
    lib.wt_debug_unit_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    n = lib.wt_debug_unit_clocks(e._h, clk, units, cap)
    c = np.array(clk[:2 * n], dtype=np.uint64).reshape(n, 2).astype(np.int64)
    u = np.array(units[:4 * n]).reshape(n, 4)
    live = (u[:, 1] - u[:, 0]) > 0
    t0 = c[live, 0].min()
    st, en = (c[live, 0] - t0) / 100.0, (c[live, 1] - t0) / 100.0          # us
    q = [0, 1, 10, 25, 50, 75, 90, 99, 100]
    print(f"{nx}x{ny} body {int(body)}: {ms / 100 * 1000:.1f} us per four-step pass (halo kernel included); {int(live.sum())} live units of {n}")
    print("  percentile        " + "  ".join(f"{v:6d}" for v in q))
    print("  unit starts, us   " + "  ".join(f"{v:6.1f}" for v in np.percentile(st, q)))
    print("  unit ends, us     " + "  ".join(f"{v:6.1f}" for v in np.percentile(en, q)))
    print("  durations, us     " + "  ".join(f"{v:6.1f}" for v in np.percentile(en - st, q)))
    # occupancy over the launch: live units per 2-us bin
    T = en.max()
    bins = np.arange(0, T + 2.0, 2.0)
    occ = [(int(((st <= b) & (en > b)).sum())) for b in bins]
    print("  units alive at t = 0, 2, 4 ... us: " + " ".join(str(v) for v in occ))
    blk = np.arange(n)[live] // 4
    print("  first start of workgroup k (dispatch order), us, every 32nd: " + " ".join(f"{st[blk == k].min():.1f}" for k in range(0, int(blk.max()) + 1, 32) if (blk == k).any()))
    # older / younger: workgroup k runs on XCD k % 8; on its XCD it is number k // 8, and with two workgroups per CU the second 32 of an XCD's 64 share their
    # CUs (and SIMDs) with the first 32.  Median duration and end of the units by that half (launch order as dispatched: odd passes run the list reversed).
    dur = en - st
    nblk = (n + 3) // 4
    for rev in (0, 1):
        pos = (nblk - 1 - blk) if rev else blk
        first = (pos // 8) < 32
        if first.any() and (~first).any():
            print(f"  assuming {'reversed' if rev else 'forward'} dispatch: first-dispatched half: duration median {np.median(dur[first]):.1f} end median {np.median(en[first]):.1f} max {en[first].max():.1f};"
                  f" second half: duration median {np.median(dur[~first]):.1f} end median {np.median(en[~first]):.1f} max {en[~first].max():.1f}")
