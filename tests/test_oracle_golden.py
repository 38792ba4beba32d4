"""Pins the CPU oracle to the reference.

tests/golden/run_*.npz were produced by oracle/make_goldens.py: the reference's OWN shader text
(STEP_FS html:222-360, RENDER_FS html:362-422) executed headless per lattice site, and its own
JS reductions (html:596-614, 650-700), with NX/NY overridden.  The NumPy transcription and the C
restatement must reproduce those outputs BIT FOR BIT (fp32 and fp64).
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits_equal


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _mask_from_spans(spans, nx, ny):
    m = np.zeros((ny, nx), np.uint8)
    for iy, a, b in spans:
        m[iy, a:b + 1] = 255
    return m


RUNS = sorted(glob.glob(os.path.join(GOLDEN, "run_*.npz")))
assert RUNS, "golden run fixtures missing"


def _replay(oracle, g, pkg):
    """Replays a golden run (incl. the AoA / U0 slider schedules) with `oracle`."""
    nx, ny, steps = int(g["nx"]), int(g["ny"]), int(g["steps"])
    dt = np.float32 if str(g["mode"]) == "f32" else np.float64
    sched = json.loads(str(g["schedules"]))
    events = sorted({0, steps} | {e["step"] for e in sched["aoa_schedule"]} | {e["step"] for e in sched["u0_schedule"]})
    mask = _mask_from_spans(g["mask0_spans"], nx, ny)
    u0, tau = float(g["u0"]), float(g["tau"])
    f, macro = oracle.equilibrium_init(nx, ny, u0, dt)
    for a, b in zip(events[:-1], events[1:]):
        for e in sched["aoa_schedule"]:
            if e["step"] == a:
                mask = pkg.geometry.build_geometry(nx, ny, e["aoa"], None, str(g["shape"])).mask
        for e in sched["u0_schedule"]:
            if e["step"] == a:
                u0 = e["u0"]
        f, macro = oracle.run(mask, b - a, tau, u0, dt, f=f)
    return f, macro, mask, u0


@pytest.mark.parametrize("path", RUNS, ids=[os.path.basename(p)[:-4] for p in RUNS])
@pytest.mark.parametrize("which", ["numpy", "c"])
def test_oracle_reproduces_reference_shader(path, which, oracle_np, oracle_c, pkg):
    g = np.load(path)
    if which == "numpy" and int(g["nx"]) * int(g["ny"]) * int(g["steps"]) > 6e6:
        pytest.skip("NumPy transcription is exercised on the small cases; C restatement covers this one")
    oracle = oracle_np if which == "numpy" else oracle_c
    f, (rho, ux, uy), _, _ = _replay(oracle, g, pkg)
    assert _sha(f) == str(g["f_sha256"])
    if "f" in g.files:
        assert bits_equal(f, g["f"])
    assert bits_equal(rho, g["rho"]) and bits_equal(ux, g["ux"]) and bits_equal(uy, g["uy"])


@pytest.mark.parametrize("path", [p for p in RUNS if "_f32" in p], ids=lambda p: os.path.basename(p)[:-4])
def test_reductions_match_reference_js(path, oracle_np, oracle_c, pkg):
    """updateFieldsFromMacro (html:596-614) and computeForces (html:650-700) run by Node on the
    golden macro field vs the oracle's restatement of them."""
    g = np.load(path)
    nx, ny = int(g["nx"]), int(g["ny"])
    _, _, mask, u0 = _replay(oracle_c, g, pkg)
    rho, ux, uy = g["rho"], g["ux"], g["uy"]
    mx, cmin, cmax = oracle_np.ranges_from_macro(rho, ux, uy, mask, u0)
    np.testing.assert_allclose([mx, cmin, cmax], g["ranges"], rtol=1e-13, atol=0)
    U, V, C = oracle_np.normalised_fields(rho, ux, uy, mask, u0)
    assert _sha(np.stack([U, V, C])) == str(g["fields_sha256"])
    fx, fy, surf, rev = oracle_np.compute_forces_raw(rho, ux, mask)
    st = oracle_np.ForceState()
    st.update(fx, fy, surf, rev, u0, nx)
    np.testing.assert_allclose([st.cl, st.cd, st.sep], g["forces_first"], rtol=1e-11, atol=1e-13)
    st.update(fx, fy, surf, rev, u0, nx)
    np.testing.assert_allclose([st.cl, st.cd, st.sep], g["forces_second"], rtol=1e-11, atol=1e-13)


def test_lattice_constants_and_init(oracle_np):
    """dir/wt/opp as the shader's functions return them (html:238-264); equilibriumInitData (html:474-490)."""
    with open(os.path.join(GOLDEN, "misc.json")) as fh:
        misc = json.load(fh)
    lat = misc["lattice"]
    assert [tuple(e) for e in lat["e"]] == list(oracle_np.E)
    assert list(lat["opp"]) == list(oracle_np.OPP)
    assert [np.float32(w) for w in lat["w"]] == list(oracle_np.weights(np.float32))
    for key, u0 in (("init_u0_0.06", 0.06), ("init_u0_0.084", 0.084)):
        f, (rho, ux, uy) = oracle_np.equilibrium_init(4, 3, u0, np.float32)
        assert [float(v) for v in f[:, 0, 0]] == misc[key]["f"]
        assert [float(rho[0, 0]), float(ux[0, 0]), float(uy[0, 0])] == misc[key]["macro"]
    c = misc["consts_320x160"]
    assert c["TAU"] == oracle_np.TAU_DEFAULT and c["VORT_SCALE"] == oracle_np.VORT_SCALE
    assert (c["DX0"], c["DX1"]) == (oracle_np.DX0, oracle_np.DX1)
    assert oracle_np.lattice_reynolds(0.06, 320, 0.58) == pytest.approx(0.06 * c["CHORD_L"] / c["NU_L"], rel=1e-14)
    assert round(oracle_np.lattice_reynolds(0.06, 320, 0.58)) == 391          # SURVEY §8c known answer


def test_known_answers_from_survey():
    """Solid-cell counts the survey captured from the reference's rasterMask."""
    want = {"geom_default_320x160_naca2412_a6": 2463, "geom_cfg1_256x128_naca0012_a0": 1566,
            "geom_cfg2_1024x512_naca2412_a5": 25283, "geom_cfg5_4096x2048_naca4412_a12": 405515,
            "geom_refaxes_4096x4096_naca0012_a10": 808217}
    for name, count in want.items():
        assert int(np.load(os.path.join(GOLDEN, name + ".npz"))["solid_count"]) == count


def test_field_scalar_vs_reference_render_shader(oracle_np):
    """RENDER_FS executed on the golden macro field: recolour the oracle's scalar t with the
    shader's colour maps (html:371-393, restated here for the test) and compare RGB."""
    g = np.load(os.path.join(GOLDEN, "run_64x32_naca0012_a0_f32.npz"))
    nx, ny = int(g["nx"]), int(g["ny"])
    mask = _mask_from_spans(g["mask0_spans"], nx, ny)
    rgb = g["render_rgb"]                       # [mode][ny][nx][3]
    mx, cmin, cmax = (float(v) for v in g["ranges"])
    F = np.float32

    def lerp_stops(t, stops):
        stops = (np.asarray(stops, dtype=F) / F(255.0)).astype(F)
        t = np.minimum(np.maximum(t, F(0)), F(1))
        n = len(stops) - 1
        f = (t * F(n)).astype(F)
        i = np.clip(np.floor(f).astype(np.int64), 0, n - 1)
        u = (f - i.astype(F)).astype(F)[..., None]
        return (stops[i] * (F(1) - u) + stops[i + 1] * u).astype(F)

    SPEED = [[5, 5, 20], [0, 20, 120], [0, 60, 200], [0, 140, 220], [0, 220, 220], [0, 210, 140], [80, 200, 0], [220, 210, 0], [255, 120, 0], [220, 20, 0]]
    CP = [[20, 50, 160], [40, 110, 210], [100, 175, 235], [190, 220, 245], [248, 248, 248], [248, 214, 140], [240, 150, 60], [205, 50, 25]]
    fluid = mask == 0
    for mode, stops in ((0, SPEED), (1, CP)):
        t = oracle_np.field_scalar(mode, g["rho"], g["ux"], g["uy"], mask, float(g["u0"]), mx, cmin, cmax)
        col = lerp_stops(t[fluid], stops)
        np.testing.assert_allclose(col, rgb[mode][fluid], rtol=0, atol=2e-6)
    t = oracle_np.field_scalar(2, g["rho"], g["ux"], g["uy"], mask, float(g["u0"]), mx, cmin, cmax)[fluid]
    t = np.clip(t, F(-1), F(1))
    base = np.asarray([0.06, 0.07, 0.11], F)
    neg, pos = np.asarray([0.15, 0.5, 0.98], F), np.asarray([0.98, 0.28, 0.18], F)
    a = np.abs(t)[..., None]
    col = np.where((t < 0)[..., None], base * (F(1) - a) + neg * a, base * (F(1) - a) + pos * a)
    np.testing.assert_allclose(col, rgb[2][fluid], rtol=0, atol=2e-6)
    np.testing.assert_allclose(rgb[0][~fluid], np.broadcast_to(np.asarray([0.039, 0.043, 0.078], F), rgb[0][~fluid].shape), atol=1e-7)
