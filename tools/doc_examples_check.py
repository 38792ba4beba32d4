"""Runs the code examples of INTEGRATION.md (sections 1 and 2) to make sure the documentation is executable."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (same HIP runtime as the package would pick)

# --- section 2: minimal ctypes binding ---
lib = ctypes.CDLL(os.path.join(ROOT, "airfoil-cfd-tool_amd/lib/libwindtunnel.so"))
lib.wt_last_error.restype = ctypes.c_char_p
h = ctypes.c_void_p()
def ok(rc):
    if rc: raise RuntimeError(lib.wt_last_error().decode())
mask = np.zeros((512, 1024), np.uint8); mask[200:300, 300:500] = 1
ok(lib.wt_create(1024, 512, 0, 0, ctypes.byref(h)))
ok(lib.wt_set_mask(h, mask.ctypes.data_as(ctypes.c_void_p)))
ok(lib.wt_init_equilibrium(h, ctypes.c_double(0.06)))
ok(lib.wt_step(h, 4, ctypes.c_double(0.58), ctypes.c_double(0.06)))
rho = np.empty((512, 1024), np.float32); ux = np.empty_like(rho); uy = np.empty_like(rho)
ok(lib.wt_read_macro(h, *(a.ctypes.data_as(ctypes.c_void_p) for a in (rho, ux, uy))))
ok(lib.wt_destroy(h))
assert np.isfinite(rho).all() and abs(float(ux.mean()) - 0.06) < 0.01

# --- section 1: host API ---
import airfoil_cfd_tool_amd as wtamd
coords_after = wtamd.geometry.SHAPES["naca2412"]()
wt = wtamd.build_lbm_component(coords_after, "NACA 2412", nx=1024, ny=512)
wt.aoa_deg = 8.0; wt.set_field("vort"); wt.set_flow_speed(0.06)
for _ in range(15):
    wt.frame(render=False)
img = wt.render_rgba()
s = wt.stats()
path = wt.save_png(os.path.join("/tmp", wt.png_name()))
assert img.shape == (512, 1024, 4) and s.cl is not None and os.path.getsize(path) > 1000
print("doc examples ok:", s, path)
wt.close()

# --- section 1, the canvas paragraph: trail layer + tracers + composited frame on the device ---
from airfoil_cfd_tool_amd.tracers import Tracers
with wtamd.WindTunnel(coords_after, "NACA 2412", nx=320, ny=160) as wt:
    layer, tr = wt.trail_layer(), Tracers(wt, seed=1)
    for _ in range(10):
        wt.frame(render=False)
        tr.draw(layer, 16.0)
    img = wt.compose_frame(trails=layer)
    assert img.shape == (360, 680, 4) and img.dtype == np.uint8
print("canvas example ok")
