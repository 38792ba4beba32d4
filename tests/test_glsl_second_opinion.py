"""A second, independent execution of the reference's shader TEXT (oracle/ref_py/glsl_simt.py: an AST interpreter in Python that runs all
lattice sites at once in NumPy binary32 / binary64) must reproduce the committed goldens — which oracle/ref_js/glsl2js.js (a transpiler to
JavaScript, one site at a time, Math.fround) produced — bit for bit.  The two interpreters share no code, language or arithmetic engine; the
goldens therefore do not rest on one hand-written GLSL semantics (parity caveat (i)).

Needs the reference's HTML (the shader text is read from it at run time, nothing of it is stored): runs in the build container, skips on
the GPU box and anywhere else /root/reference is absent."""
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT

REF_HTML = "/root/reference/pages/airfoil_flow_lbm_aerolab.html"
pytestmark = pytest.mark.skipif(not os.path.exists(REF_HTML), reason="the reference's HTML is not present here")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def _shader_text(name):
    txt = open(REF_HTML, encoding="utf-8").read()
    m = re.search(r"const\s+" + name + r"\s*=\s*`(.*?)`;", txt, flags=re.S)
    assert m, name
    return m.group(1)


def _mask_from_spans(spans, nx, ny):
    mask = np.zeros((ny, nx), np.uint8)
    for iy, x0, x1 in spans:
        mask[iy, x0:x1 + 1] = 255
    return mask


def _run_steps(g, steps, mask=None, hooks=None):
    from ref_py.glsl_simt import Sampler, Shader, Vec
    nx, ny = int(g["nx"]), int(g["ny"])
    dt = np.float32 if str(g["mode"]) == "f32" else np.float64
    sh = Shader(_shader_text("STEP_FS_SRC"), dt)
    if mask is None:
        mask = _mask_from_spans(g["mask0_spans"], nx, ny)
    n = nx * ny
    init = np.asarray(g["init_f"], dtype=np.float64).astype(dt)          # equilibriumInitData: doubles rounded to the storage type
    A = np.empty((ny, nx, 4), dt); B = np.empty((ny, nx, 4), dt); C = np.empty((ny, nx, 4), dt)
    A[...] = init[0:4]; B[...] = init[4:8]
    C[..., 0] = init[8]; C[..., 1] = dt(1.0); C[..., 2] = dt(float(g["u0"])); C[..., 3] = dt(0.0)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny))
    vUV = Vec([((ix.ravel() + 0.5) / nx).astype(dt), ((iy.ravel() + 0.5) / ny).astype(dt)])
    uni = dict(texel=Vec([dt(1.0 / nx), dt(1.0 / ny)]), gridSize=Vec([nx, ny], True), tau=dt(float(g["tau"])), U0=dt(float(g["u0"])), vUV=vUV)
    for s in range(steps):
        if hooks:
            hooks(s, uni)
        out = sh.run(n, texA=Sampler(A), texB=Sampler(B), texC=Sampler(C), texMask=Sampler(mask), **uni)
        A = np.stack(out["outA"].c, axis=-1).reshape(ny, nx, 4).astype(dt)
        B = np.stack(out["outB"].c, axis=-1).reshape(ny, nx, 4).astype(dt)
        C = np.stack(out["outC"].c, axis=-1).reshape(ny, nx, 4).astype(dt)
    f = np.concatenate([np.moveaxis(A, -1, 0), np.moveaxis(B, -1, 0), C[None, ..., 0]], axis=0)
    return f, C[..., 1], C[..., 2], C[..., 3], C


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("name", ["run_64x32_naca0012_a0_f32", "run_64x32_naca0012_a0_f64", "run_96x48_naca4412_a20_lowtau_f32"])
def test_second_interpreter_reproduces_the_step_goldens(name):
    """STEP_FS, all branches: solid, outlet, far field, interior with bounce-back; the low-tau run drives the stability clamp."""
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    f, rho, ux, uy, _ = _run_steps(g, int(g["steps"]))
    assert _same_bits(f, g["f"]), float(np.abs(f.astype(np.float64) - g["f"]).max())
    assert _same_bits(rho, g["rho"]) and _same_bits(ux, g["ux"]) and _same_bits(uy, g["uy"])


def test_second_interpreter_reproduces_the_render_golden():
    """RENDER_FS on the golden macro texture, the three field modes: the floats the colour maps produce, before the GL quantisation."""
    from ref_py.glsl_simt import Sampler, Shader, Vec
    g = np.load(os.path.join(GOLD, "run_64x32_naca0012_a0_f32.npz"), allow_pickle=True)
    if "render_rgb" not in g.files:
        pytest.skip("no render golden in this file")
    nx, ny = int(g["nx"]), int(g["ny"])
    dt = np.float32
    _, _, _, _, C = _run_steps(g, int(g["steps"]))
    sh = Shader(_shader_text("RENDER_FS_SRC"), dt)
    mask = _mask_from_spans(g["mask0_spans"], nx, ny)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny))
    vUV = Vec([((ix.ravel() + 0.5) / nx).astype(dt), ((iy.ravel() + 0.5) / ny).astype(dt)])
    max_s, cp_min, cp_max = (float(v) for v in g["ranges"])
    for mode in range(3):
        out = sh.run(nx * ny, texC=Sampler(C), texMask=Sampler(mask), texel=Vec([dt(1.0 / nx), dt(1.0 / ny)]), fieldMode=mode, U0=dt(float(g["u0"])),
                     maxS=dt(max_s), cpMin=dt(cp_min), cpMax=dt(cp_max), vortScale=dt(0.06), vUV=vUV)
        rgb = np.stack(out["fragColor"].c[:3], axis=-1).reshape(ny, nx, 3).astype(dt)
        assert _same_bits(rgb, np.asarray(g["render_rgb"][mode], dtype=dt)), (mode, float(np.abs(rgb - g["render_rgb"][mode]).max()))
