"""GPU: the product (WindTunnel -> C-ABI -> HIP kernels) against the committed golden vectors
(reference shader + reference JS run under Node) and against the oracle's reductions."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits_equal

pytestmark = pytest.mark.gpu

RUNS_F32 = sorted(glob.glob(os.path.join(GOLDEN, "run_*_f32.npz")))
RUNS_ALL = sorted(glob.glob(os.path.join(GOLDEN, "run_*.npz")))


def _drive(pkg, g, options=None):
    """Drives a WindTunnel through a golden run with the page's controls (AoA slider, U0 slider).
    `options`: engine options set before the first step (forces one kernel form onto the small golden lattices)."""
    nx, ny, steps = int(g["nx"]), int(g["ny"]), int(g["steps"])
    sched = json.loads(str(g["schedules"]))
    events = sorted({0, steps} | {e["step"] for e in sched["aoa_schedule"]} | {e["step"] for e in sched["u0_schedule"]})
    wt = pkg.WindTunnel(shape=str(g["shape"]), nx=nx, ny=ny, aoa_deg=float(g["aoa"]), u0=float(g["u0"]),
                        tau=float(g["tau"]), dtype="float32" if str(g["mode"]) == "f32" else "float64")
    for name, value in (options or ()):
        wt.engine.set_option(name, value)
    for a, b in zip(events[:-1], events[1:]):
        for e in sched["aoa_schedule"]:
            if e["step"] == a:
                wt.aoa_deg = e["aoa"]
        for e in sched["u0_schedule"]:
            if e["step"] == a:
                wt.set_flow_speed(e["u0"])
        wt.sim_step(b - a)
    return wt


# every kernel form meets the fixtures generated from html:283-360 directly: the single-step kernel (depth 0) and the marching
# kernels with 2 / 3 / 4 steps per pass forced onto these small lattices (fuse_steps = 2), each with the proved fast division by
# tau and with the IEEE one
# ... and, round 5, the three- and four-step kernels on OVERLAPPING windows (window_overlap = 1: margins of four rows in place of the halo lines; fp32 — an
# fp64 handle keeps windows that tile the column whatever the option says)
FORMS = [(d, fd, 0) for d in (0, 2, 3, 4) for fd in (1, 0)] + [(d, fd, 1) for d in (3, 4) for fd in (1, 0)]


@pytest.mark.parametrize("depth,fast_div,overlap", FORMS, ids=[f"depth{d}-fd{fd}" + ("-overlap" if o else "") for d, fd, o in FORMS])
@pytest.mark.parametrize("path", RUNS_ALL, ids=[os.path.basename(p)[:-4] for p in RUNS_ALL])
def test_gpu_reproduces_reference_shader_goldens(pkg, path, depth, fast_div, overlap):
    g = np.load(path)
    opts = [("fast_div", fast_div)]
    if depth:
        opts += [("fuse_depth", depth), ("fuse_steps", 2), ("window_overlap", overlap)]
    else:
        opts += [("fuse_steps", 0)]
    with _drive(pkg, g, opts) as wt:
        if depth:       # the forced plan must really be the one that ran (mask changes mid-run rebuild it in place)
            assert wt.engine.get_option("fuse_active") == 1.0 and wt.engine.get_option("fuse_depth") == depth
            assert wt.engine.get_option("single_steps") <= int(g["steps"]) // 2
            if depth >= 3:
                assert wt.engine.get_option("window_overlap") == (1.0 if overlap and str(g["mode"]) == "f32" else 0.0)
        rho, ux, uy = wt.read_macro()
        f = wt.read_f()
    # stated fp tolerance ...
    assert np.abs(rho.astype(np.float64) - g["rho"]).max() <= 1e-5
    assert max(np.abs(ux.astype(np.float64) - g["ux"]).max(), np.abs(uy.astype(np.float64) - g["uy"]).max()) <= 5e-6
    # ... and what is actually achieved: bit-exact
    assert bits_equal(rho, g["rho"]) and bits_equal(ux, g["ux"]) and bits_equal(uy, g["uy"])
    if "f" in g.files:
        assert bits_equal(f, g["f"])


@pytest.mark.parametrize("path", RUNS_F32, ids=[os.path.basename(p)[:-4] for p in RUNS_F32])
def test_gpu_reductions_vs_reference_js(pkg, path):
    """wt_reduce_ranges / wt_forces + the host's smoothing vs updateFieldsFromMacro / computeForces
    as Node ran them (html:596-614, 650-700)."""
    g = np.load(path)
    with _drive(pkg, g) as wt:
        got = wt.update_fields_from_macro()
        np.testing.assert_allclose(got, g["ranges"], rtol=1e-12, atol=0)
        wt.compute_forces()
        np.testing.assert_allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], g["forces_first"], rtol=1e-10, atol=1e-12)
        wt.compute_forces()
        np.testing.assert_allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], g["forces_second"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_gpu_field_scalars_vs_oracle(pkg, oracle_np, dtype):
    nx, ny = 512, 256
    with pkg.WindTunnel(shape="naca4412", nx=nx, ny=ny, aoa_deg=14.0, dtype=dtype) as wt:
        wt.sim_step(150)
        rho, ux, uy = wt.read_macro()
        mx, cmin, cmax = wt.update_fields_from_macro()
        ref_ranges = oracle_np.ranges_from_macro(rho, ux, uy, wt.geometry.mask, wt.u0)
        np.testing.assert_allclose((mx, cmin, cmax), ref_ranges, rtol=1e-13)
        for name, mode in (("speed", 0), ("cp", 1), ("vort", 2)):
            t = wt.render_field(field=name)
            ref = oracle_np.field_scalar(mode, rho, ux, uy, wt.geometry.mask, wt.u0, mx, cmin, cmax)
            assert t.dtype == ref.dtype and np.array_equal(np.isnan(t), np.isnan(ref))
            assert bits_equal(np.nan_to_num(t), np.nan_to_num(ref)), name
        fx, fy, surf, rev = wt.engine.forces()
        rfx, rfy, rsurf, rrev = oracle_np.compute_forces_raw(rho, ux, wt.geometry.mask)
        assert (surf, rev) == (rsurf, rrev)
        np.testing.assert_allclose([fx, fy], [rfx, rfy], rtol=1e-11, atol=1e-12)


def test_ranges_keep_previous_when_nothing_qualifies(pkg):
    """html:611-613: with no fluid cell at all the ranges keep their initial values (html:593)."""
    nx = ny = 64
    with pkg.WindTunnel(nx=nx, ny=ny) as wt:
        wt.engine.set_mask(np.ones((ny, nx), np.uint8))
        wt.sim_step(2)
        assert wt.update_fields_from_macro() == (0.6, -1.0, 1.0)
        assert wt.compute_forces() is None and wt.cl_smooth is None


def test_frame_loop_and_stats(pkg, oracle_c, oracle_np):
    """frame() (html:902-930): 4 steps, render with the previous ranges, range update, forces every 3rd frame."""
    nx, ny = 320, 160
    with pkg.WindTunnel(nx=nx, ny=ny) as wt:            # reference defaults: NACA 2412, 6 deg, U0 0.06
        prev = (wt.max_s, wt.cp_min, wt.cp_max)
        st = oracle_np.ForceState()
        f = None
        for frame in range(1, 7):
            t = wt.frame()
            f, (rho, ux, uy) = oracle_c.run(wt.geometry.mask, 4, 0.58, 0.06, np.float32, f=f)
            ref_t = oracle_np.field_scalar(0, rho, ux, uy, wt.geometry.mask, 0.06, *prev)
            assert bits_equal(np.nan_to_num(t), np.nan_to_num(ref_t))
            prev = oracle_np.ranges_from_macro(rho, ux, uy, wt.geometry.mask, 0.06, prev)
            np.testing.assert_allclose((wt.max_s, wt.cp_min, wt.cp_max), prev, rtol=1e-13)
            if frame % 3 == 0:
                st.update(*oracle_np.compute_forces_raw(rho, ux, wt.geometry.mask), 0.06, nx)
        s = wt.stats()
        np.testing.assert_allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], [st.cl, st.cd, st.sep], rtol=1e-10, atol=1e-12)
        assert round(s.reynolds) == 391 and s.separation == oracle_np.stall_label(st.sep)
        assert wt.png_name() == "Uploaded_airfoil_alpha6.0deg_lbm.png"


def test_build_lbm_component_with_user_coords(pkg, oracle_c):
    """The drop-in boundary (pages/Airfoil_Analysis.py:20): coords_after + name in, running tunnel out."""
    g = np.load(os.path.join(GOLDEN, "geom_user_open_te_320x160_a4.npz"))
    coords = [[float(x) + 1e-9, float(y)] for x, y in g["user_coords"]]      # un-rounded input
    with pkg.build_lbm_component(coords, "NACA 0012 (UIUC)", aoa_deg=4.0) as wt:
        assert int((wt.geometry.mask != 0).sum()) == int(g["solid_count"])
        wt.sim_step(40)
        f = wt.read_f()
        assert wt.png_name() == "NACA_0012_(UIUC)_alpha4.0deg_lbm.png"
        f_ref, _ = oracle_c.run(wt.geometry.mask, 40, 0.58, 0.06, np.float32)
        assert bits_equal(f, f_ref)


def test_render_rgba_vs_reference_render_shader(pkg, tmp_path):
    """wt_render_rgba against RENDER_FS (html:362-422) run by Node on the golden macro field: the
    shader's float colours quantised like a GL RGBA8 framebuffer (round to nearest)."""
    g = np.load(os.path.join(GOLDEN, "run_64x32_naca0012_a0_f32.npz"))
    want = np.floor(g["render_rgb"].astype(np.float64) * 255.0 + 0.5).astype(np.uint8)     # [mode][ny][nx][3]
    with _drive(pkg, g) as wt:
        wt.update_fields_from_macro()
        for mode, name in enumerate(("speed", "cp", "vort")):
            img = wt.render_rgba(name)
            assert img.shape == (32, 64, 4) and img.dtype == np.uint8 and (img[..., 3] == 255).all()
            assert np.array_equal(img[..., :3], want[mode]), name
        path = wt.save_png(str(tmp_path / "x.png"), "speed", composite=False)      # the bare lattice field; composited canvas: tests/test_compose.py
        data = open(path, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n" and b"IHDR" in data[:32] and data[-8:-4] == b"IEND"
        import struct
        assert struct.unpack(">II", data[16:24]) == (64, 32)


def test_render_rgba_fp64_and_slabs(pkg):
    nx, ny = 512, 256
    mask = pkg.geometry.build_geometry(nx, ny, 10.0, None, "naca2412").mask
    with pkg.Engine(nx, ny, dtype="float64") as e:
        e.set_mask(mask); e.init_equilibrium(0.06); e.step(60, 0.58, 0.06)
        rng = e.reduce_ranges(0.06)
        want = [e.render_rgba(m, 0.06, rng[0], rng[1], rng[2], 0.06) for m in range(3)]
        assert (want[0][mask != 0][:, :3] == np.array([10, 11, 20], np.uint8)).all()      # solid colour, html:397
    es = [pkg.Engine(nx, ny, dtype="float64", rank=r, nranks=2, halo=2) for r in range(2)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.init_equilibrium(0.06)
        pkg.Engine.step_group(es, 60, 0.58, 0.06)
        for m in range(3):
            got = np.concatenate([e.render_rgba(m, 0.06, rng[0], rng[1], rng[2], 0.06) for e in es], axis=1)
            assert np.array_equal(got, want[m])
    finally:
        for e in es:
            e.close()


def test_from_dat_upload_path(pkg, oracle_c, tmp_path):
    """.dat upload -> parser repairs -> coords_after -> tunnel (main.py:543-608 -> AA.py:1413 -> html:561)."""
    import json
    with open(os.path.join(GOLDEN, "datfile_cases.json"), encoding="utf-8") as fh:
        case = next(c for c in json.load(fh)["cases"] if c["name"] == "lednicer_counts_header")
    p = tmp_path / "my foil.dat"
    p.write_text(case["text"])
    with pkg.WindTunnel.from_dat(str(p), nx=256, ny=128, aoa_deg=5.0) as wt:
        assert wt.parser_fixes == case["fixes"] and wt.name == "my foil"
        assert wt.png_name() == "my_foil_alpha5.0deg_lbm.png"
        wt.sim_step(50)
        f_ref, _ = oracle_c.run(wt.geometry.mask, 50, 0.58, 0.06, np.float32)
        assert bits_equal(wt.read_f(), f_ref)
