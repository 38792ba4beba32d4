"""Developer check: chain blocks against solo units on one lattice; reports where the populations differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import airfoil_cfd_tool_amd as pkg
nx, ny, depth, nsteps = (int(v) for v in sys.argv[1:5])
body = len(sys.argv) > 5
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask if body else np.zeros((ny, nx), np.uint8)
out = []
for chain in (0, 1):
    with pkg.Engine(nx, ny) as e:
        e.set_option("chain", chain); e.set_option("fuse_depth", depth); e.set_option("fuse_steps", 2)
        e.set_mask(mask); e.init_equilibrium(0.06)
        # a non-uniform start so that every column matters
        f = e.read_f()
        rng = np.random.default_rng(1)
        f *= (1 + 1e-3 * rng.standard_normal(f.shape)).astype(f.dtype)
        e.write_f(f)
        e.step(nsteps, 0.58, 0.06)
        out.append(e.read_f())
        print("chain", chain, "units", e.get_option("fuse_units"), "chain units", e.get_option("chain_units"), "passes", e.get_option("passes"))
d = (out[0].view(np.uint32) != out[1].view(np.uint32)).any(axis=0)
print("differing sites:", int(d.sum()))
if d.any():
    cols = np.where(d.any(axis=0))[0]; rows = np.where(d.any(axis=1))[0]
    print("columns:", cols[:40], "...", cols[-10:], "count", len(cols))
    print("rows:", rows[:40], "...", rows[-10:], "count", len(rows))
