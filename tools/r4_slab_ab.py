"""Runs ON THE GPU BOX: stand-alone cost of chosen slabs of the equal 8-way split of the bench tunnel under a build of the library (WT_AB_LIB):
python tools/r4_slab_ab.py [rank ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import airfoil_cfd_tool_amd._capi as capi
if os.environ.get("WT_AB_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", os.environ["WT_AB_LIB"])
    capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
nx = ny = 4096
mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca6409").mask
edges = pkg.slab_edges(nx, 8)
ranks = [int(a) for a in sys.argv[1:]] or [0, 2, 3, 4]
print(os.environ.get("WT_AB_LIB", "in-tree"), " ".join(f"slab {r}: {pkg.measure_slab_cost(mask, edges, r, 16, steps=408, trimmed=False):.2f}" for r in ranks), "us per step", flush=True)
