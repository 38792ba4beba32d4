"""Runs ON THE GPU BOX: what the measured cut costs per new mask — the first stepping call after wt_set_mask against the next one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg
for nx, ny, dtype in ((4096, 4096, "float32"), (1024, 512, "float32"), (2048, 1024, "float32"), (4096, 2048, "float64")):
    masks = [pkg.geometry.build_geometry(nx, ny, a, None, "naca2412").mask for a in (4.0, 5.0, 6.0, 7.0)]
    for tune in (1, 0):
        with pkg.Engine(nx, ny, dtype=dtype) as e:
            e.set_option("tune", tune)
            e.set_mask(masks[0]); e.init_equilibrium(0.06); e.step(8, 0.58, 0.06); e.sync()
            first, nxt = [], []
            for m in masks[1:] + masks[:1]:
                e.step(64, 0.58, 0.06)                 # (a mask that follows a short-lived one is not timed at once)
                e.set_mask(m); e.sync()
                t0 = time.perf_counter(); e.step(4, 0.58, 0.06); e.sync(); first.append((time.perf_counter() - t0) * 1e3)
                t0 = time.perf_counter(); e.step(4, 0.58, 0.06); e.sync(); nxt.append((time.perf_counter() - t0) * 1e3)
            print(f"{nx}x{ny} {dtype} tune={tune}: first step(4) after a new mask {sorted(first)[len(first) // 2]:.2f} ms, the next step(4) {sorted(nxt)[len(nxt) // 2]:.2f} ms"
                  f" (rounds {int(e.get_option('tune_rounds'))}, depth {int(e.get_option('fuse_depth')) if e.get_option('fuse_active') else 0})", flush=True)
