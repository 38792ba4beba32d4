"""Runs ON THE GPU BOX under `rocprofv3 --kernel-trace`: 200 steps of a 544-column slab-sized tunnel and of the bench lattice, so that the trace
holds the per-pass kernels (k_halo4, k_march3) of both; tools/rocpd_summary.py condenses it.  argv: widths (default 544 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd._capi as capi
if os.environ.get("WT_AB_LIB"):          # another build of the library (tools/ab/)
    capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ["WT_AB_LIB"])
    capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
for nx in [int(a) for a in sys.argv[1:]] or [544, 4096]:
    ny = 4096
    mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
    with pkg.Engine(nx, ny) as e:
        if nx < 4096:
            e.set_option("plan_columns", nx)
        e.set_mask(mask); e.init_equilibrium(0.06); e.step(200, 0.58, 0.06)
        ms = e.step_timed(400, 0.58, 0.06) / 400
        print(f"{nx}x{ny}: {ms * 1e3:.2f} us per step, depth {int(e.get_option('pass_depth'))}, units {int(e.get_option('fuse_units'))}", flush=True)
