"""Runs ON THE GPU BOX: pipelined passes (option "pipeline" = k ranges) on the bench lattice and BASELINE configs[4]: per-step time for several k, and the
populations against the un-pipelined run (bit for bit).   python tools/r5_pipeline.py [k ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg

ks = [int(a) for a in sys.argv[1:]] or [0, 2, 3, 4, 6, 8]
for nx, ny, dtype, shape, aoa, tau in ((4096, 4096, "float32", "naca6409", 10.0, 0.58), (4096, 2048, "float64", "naca4412", 12.0, 0.5 + 3 * 0.06 * (4096 / 1.84) / 1e6)):
    mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
    ref = None
    for rep in range(2):
        for k in ks:
            with pkg.Engine(nx, ny, dtype=dtype) as e:
                e.set_option("pipeline", k)
                e.set_mask(mask); e.init_equilibrium(0.06)
                e.step(203, tau, 0.06)                       # 50 four-step passes + a three-step one; the plan is tuned here
                f = e.read_f() if rep == 0 else None
                e.step(200, tau, 0.06)
                ms = e.step_timed(400, tau, 0.06) / 400
                pp = int(e.get_option("pipeline_passes"))
            same = ""
            if rep == 0:
                if ref is None:
                    ref = f
                else:
                    same = "  populations after 203 steps == un-pipelined: " + str(bool(np.array_equal(f.view(np.uint8), ref.view(np.uint8))))
            print(f"{nx}x{ny} {dtype} pipeline {k}: {ms * 1e3:7.2f} us per step = {nx * ny / ms / 1e3:8.0f} MLUPS  ({pp} pipelined passes){same}", flush=True)
