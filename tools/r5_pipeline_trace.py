"""Runs ON THE GPU BOX under rocprofv3 --kernel-trace: a few pipelined passes of the bench lattice; tools/r5_pipeline_trace.sh prints their timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nx = ny = 4096
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
with pkg.Engine(nx, ny) as e:
    e.set_option("pipeline", k); e.set_option("tune", 0)
    e.set_mask(mask); e.init_equilibrium(0.06)
    e.step(120, 0.58, 0.06); e.sync()
    e.step(40, 0.58, 0.06); e.sync()
