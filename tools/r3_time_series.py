"""Runs ON THE GPU BOX: us per step over time on the bench lattice — successive step_timed(40) calls from a cold start: how long until the rate settles?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
with pkg.Engine(nx, ny) as e:
    e.set_mask(mask); e.init_equilibrium(0.06); e.step(8, 0.58, 0.06)
    t0 = time.perf_counter(); series = []
    for k in range(120):
        series.append((time.perf_counter() - t0, e.step_timed(40, 0.58, 0.06) / 40 * 1e3))
    print("t [ms] : us/step")
    print("  ".join(f"{t * 1e3:.0f}:{u:.1f}" for t, u in series))
    time.sleep(2.0)
    print("after 2 s idle:", "  ".join(f"{e.step_timed(40, 0.58, 0.06) / 40 * 1e3:.1f}" for _ in range(12)))
