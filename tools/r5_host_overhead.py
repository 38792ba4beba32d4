"""Developer experiment: wall clock of wt_step_timed(20) against the device time it returns (whole 4096^2 lattice), and what the host calls around it cost.
    python3 tools/r5_host_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import airfoil_cfd_tool_amd as pkg
torch.cuda.set_device(0)
mask = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
with pkg.Engine(4096, 4096) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(800, 0.58, 0.06); e.sync()
    rows = []
    for _ in range(12):
        e.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter(); dev = e.step_timed(20, 0.58, 0.06); t1 = time.perf_counter()
        e.sync(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        rows.append((dev * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    a = np.array(rows[2:])
    print("step_timed(20): device us %.1f  wall of the call us %.1f  (+%.1f)   eng.sync() after it %.1f us   torch.cuda.synchronize() %.1f us" %
          (a[:, 0].mean(), a[:, 1].mean(), (a[:, 1] - a[:, 0]).mean(), a[:, 2].mean(), a[:, 3].mean()))
    rows = []
    for _ in range(12):
        e.sync()
        t0 = time.perf_counter(); e.step(20, 0.58, 0.06); t1 = time.perf_counter(); e.sync(); t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    a = np.array(rows[2:])
    print("step(20) + sync: the call returns after %.1f us, synchronised after %.1f us" % (a[:, 0].mean(), a[:, 1].mean()))
