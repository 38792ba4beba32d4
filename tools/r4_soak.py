"""Runs ON THE GPU BOX: a long run of the marching kernels against the one-step kernel on the bench lattice and on the fp64 configuration — the same
bits after thousands of steps of a developing flow (separation, vortex shedding, the clamp of html:344-350 if it comes to that), not only after the
tens of steps the parity tests take.   python tools/r4_soak.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for (nx, ny, dtype, shape, aoa, tau) in ((4096, 4096, "float32", "naca6409", 10.0, 0.58), (4096, 2048, "float64", "naca4412", 12.0, 0.5004007), (1024, 512, "float32", "naca2412", 5.0, 0.52)):
    mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
    out = []
    for fuse in (1, 0):
        with pkg.Engine(nx, ny, dtype=dtype) as e:
            e.set_option("fuse_steps", fuse)
            e.set_mask(mask); e.init_equilibrium(0.08)
            t0 = time.time()
            done = 0
            for chunk in (steps // 3, steps // 3 + 1, steps - 2 * (steps // 3) - 1):      # uneven calls: remainders of every kind
                e.step(chunk, tau, 0.08); done += chunk
            e.sync()
            f = e.read_f(); rho, ux, uy = e.read_macro()
            out.append((f, rho, ux, uy, int(e.get_option("pass_depth")), int(e.get_option("single_steps")), time.time() - t0, e.clamp_events() if hasattr(e, "clamp_events") else None))
    same = all(np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)) for a, b in zip(out[0][:4], out[1][:4]))
    finite = bool(np.isfinite(out[0][0]).all())
    print(f"{shape} {nx}x{ny} {dtype} tau {tau}: {steps} steps, marching kernels (depth {out[0][4]}, {out[0][5]} single steps, {out[0][6]:.1f} s) vs k_step ({out[1][6]:.1f} s): "
          f"populations and (rho, ux, uy) {'BIT-IDENTICAL' if same else 'DIFFER'}; finite {finite}; max |u| {float(np.nanmax(np.hypot(out[0][2], out[0][3]))):.4f}", flush=True)
