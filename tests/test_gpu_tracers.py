"""GPU: tracer advection (wt_advect_tracers) against the reference's own advect()/sampleUV()
(html:616-639, 758-771) run by Node on the golden macro fields; seeding statistics of the host
particle system (html:730-753)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["run_64x32_naca0012_a0_f32", "run_default_320x160_naca2412_a6_f32"])
def test_advect_matches_reference_js(pkg, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    with pkg.WindTunnel(shape=str(g["shape"]), nx=int(g["nx"]), ny=int(g["ny"]), aoa_deg=float(g["aoa"])) as wt:
        wt.sim_step(int(g["steps"]))
        pts, want = g["tracer_points"], g["tracer_advect"]
        xn, yn, sp, ok = wt.engine.advect_tracers(pts[:, 0], pts[:, 1], float(g["tracer_dt"]), wt.u0, (-0.42, 1.42, -0.46, 0.46))
        null = np.isnan(want[:, 0])
        assert np.array_equal(~ok, null) and null.sum() > 50 and (~null).sum() > 500
        np.testing.assert_allclose(np.stack([xn, yn, sp], 1)[ok], want[~null], rtol=1e-12, atol=1e-14)
        assert np.array_equal(xn[~ok], pts[~ok, 0])


def test_particle_system_statistics(pkg):
    """Seeding mirrors initParts/spawn: lanes span the window, 35 % centre band, ages uniform; a few
    hundred frames keep every particle inside the window and the population constant."""
    with pkg.WindTunnel(nx=320, ny=160) as wt:
        wt.sim_step(200)
        wt.update_fields_from_macro()
        tr = pkg.Tracers(wt, n=2600, seed=7)
        assert tr.x.size == 2600 and tr.x.min() >= -0.42 and tr.x.max() <= -0.42 + 1.84 * 0.95
        centre = np.abs(tr.lane) < 0.92 / 6
        assert 0.50 < centre.mean() < 0.62            # 35 % forced + a third of the uniform 65 %
        assert 0 < tr.life.min() and tr.life.max() <= 520 and 150 < tr.life.mean() < 220
        moved = 0
        for _ in range(120):
            seg, t = tr.step(16.0)
            moved += len(seg)
            assert seg.shape[1] == 4 and ((t >= 0) & (t <= 1)).all()
            assert tr.x.size == 2600 and np.isfinite(tr.x).all() and np.isfinite(tr.y).all()
            assert tr.x.min() >= -0.42 - 1e-9 and tr.x.max() <= 1.42 + 0.06 and np.abs(tr.y).max() <= 0.46 + 0.06
        assert moved > 0.9 * 120 * 2600
        # free-stream particles drift downstream at ~U0-normalised speed 1: 0.00105*16 per frame
        tr.resize(800); assert tr.x.size == 800
        tr.resize(3000); assert tr.x.size == 3000


class _FakeStreamlit:
    """Just enough of the streamlit API for build_lbm_component (streamlit is not installed here)."""

    def __init__(self):
        self.session_state = {}
        self.calls = []

    def slider(self, label, lo, hi, value, step):
        self.calls.append(("slider", label))
        return {"Angle of attack": 8.0}.get(label, value)

    def selectbox(self, label, options):
        self.calls.append(("selectbox", label))
        return options[1]

    def image(self, img, caption=None, use_column_width=None):
        self.calls.append(("image", img.shape, img.dtype, img.copy()))

    def metric(self, label, value):
        self.calls.append(("metric", label, value))

    def empty(self):            # a placeholder is updated in place: same drawing calls
        return self

    def columns(self, n):
        return [self for _ in range(n)]

    def error(self, msg):
        self.calls.append(("error", msg))


def test_streamlit_page_streams_frames_without_a_rerun(pkg):
    """f4: one call of build_lbm_component keeps the canvas advancing (the page's rAF loop, html:902-930) — several
    composited images and several read-out updates per call, the tunnel survives the rerun a widget change causes."""
    from airfoil_cfd_tool_amd.streamlit_page import build_lbm_component
    st = _FakeStreamlit()
    coords = pkg.geometry.SHAPES["naca4412"]()
    wt = build_lbm_component(coords, "NACA 4412", nx=256, ny=128, frames=12, frames_per_update=4, st=st)
    try:
        assert wt is not None and wt.aoa_deg == 8.0 and wt.field == "cp" and wt.steps == 48          # 12 frames x 4 steps
        images = [c for c in st.calls if c[0] == "image"]
        assert len(images) == 3 and all(c[1] == (360, 680, 4) and c[2] == np.dtype(np.uint8) for c in images)
        assert (images[0][3] != images[-1][3]).any()                                                # the picture moved on
        metrics = [c for c in st.calls if c[0] == "metric"]
        assert len(metrics) == 4 * 3 and {c[1] for c in metrics} == {"CL (approx)", "CD (approx)", "Reynolds", "Separation"}
        assert metrics[2][2] == "313"
        again = build_lbm_component(coords, "NACA 4412", nx=256, ny=128, frames=4, frames_per_update=2, st=st)
        assert again is wt and wt.steps == 64 and st.session_state["wt_amd_frames"] == 16          # same session -> same tunnel keeps running
        assert len([c for c in st.calls if c[0] == "image"]) == 5
    finally:
        wt.close()


class _StopScript(Exception):
    """What Streamlit does to a running script when a widget changes or the session ends."""


def test_streamlit_page_never_stops_by_itself_and_picks_up_control_changes(pkg):
    """f4, the stream without an end (html:902-930: the rAF loop never stops).  (1) A Streamlit without st.fragment: the call keeps
    streaming until Streamlit itself stops the script — here after 7 canvas updates; with `frames` unset it would otherwise never
    return.  (2) A Streamlit with st.fragment: the controls and the canvas live in a fragment that Streamlit re-runs on its timer;
    a slider moved between two runs reaches the tunnel at the next update."""
    from airfoil_cfd_tool_amd.streamlit_page import build_lbm_component
    coords = pkg.geometry.SHAPES["naca2412"]()

    class Old(_FakeStreamlit):
        def image(self, img, caption=None, use_column_width=None):
            super().image(img, caption, use_column_width)
            if len([c for c in self.calls if c[0] == "image"]) == 7:
                raise _StopScript()

    st = Old()
    with pytest.raises(_StopScript):
        build_lbm_component(coords, "NACA 2412", nx=256, ny=128, frames_per_update=2, st=st)
    wt = st.session_state["wt_amd_tunnel"]
    try:
        assert wt.steps == 7 * 2 * 4 and st.session_state["wt_amd_frames"] == 14      # 7 updates went out; nothing but the stop ended the stream

        class New(_FakeStreamlit):
            def __init__(self, session):
                super().__init__()
                self.session_state = session
                self.aoa = 8.0
                self.timer_runs = 0

            def slider(self, label, lo, hi, value, step):
                return self.aoa if label == "Angle of attack" else value

            def fragment(self, run_every=None):
                assert run_every and run_every > 0
                def deco(fn):
                    def run_by_timer():
                        for k in range(5):                        # Streamlit's timer: the fragment alone re-runs, again and again
                            if k == 3:
                                self.aoa = 11.5                   # the user moves the slider between two runs
                            fn()
                            self.timer_runs += 1
                    return run_by_timer
                return deco

        st2 = New(st.session_state)
        again = build_lbm_component(coords, "NACA 2412", nx=256, ny=128, frames_per_update=1, st=st2)
        assert again is wt and st2.timer_runs == 5 and wt.aoa_deg == 11.5
        assert len([c for c in st2.calls if c[0] == "image"]) == 5 and wt.steps == 7 * 2 * 4 + 5 * 4
    finally:
        wt.close()


def test_composited_png_of_a_running_tunnel(pkg, tmp_path):
    """f2 end to end: field from the GPU colour-map kernel, tracer strokes from the GPU advection, compositor on the
    host, PNG under the page's file name (html:980-1000)."""
    import struct
    from airfoil_cfd_tool_amd.compose import TrailLayer
    from airfoil_cfd_tool_amd.tracers import Tracers
    with pkg.WindTunnel(shape="naca2412", nx=320, ny=160, aoa_deg=6.0, name="NACA 2412 test") as wt:
        tr, layer = Tracers(wt, n=600, seed=3), TrailLayer(1)
        for _ in range(20):
            wt.frame(render=False)
            tr.draw(layer, 16.0)
        img = wt.compose_frame(trails=layer)
        assert img.shape == (360, 680, 4)
        assert (layer.a > 0.3).sum() > 300                                      # strokes exist
        # the body is drawn over field and strokes: the lattice's solid cells map to foil-coloured pixels
        cx = 54 + int((0.3 - (-0.42)) / 1.84 * 584)
        assert ((img[:, cx, :3] == np.array([0x0d, 0x10, 0x18])).all(axis=1)).sum() >= 4
        path = wt.save_png(str(tmp_path / wt.png_name()), trails=layer)
        assert path.endswith("NACA_2412_test_alpha6.0deg_lbm.png")
        raw = open(path, "rb").read()
        assert struct.unpack(">II", raw[16:24]) == (680, 360)
        bare = wt.save_png(str(tmp_path / "bare.png"), composite=False)
        assert struct.unpack(">II", open(bare, "rb").read()[16:24]) == (320, 160)


@pytest.mark.parametrize("dtype,nx,ny,field,scale", [("float32", 320, 160, "speed", 1), ("float32", 1024, 512, "vort", 1),
                                                     ("float64", 640, 320, "cp", 1), ("float32", 320, 160, "speed", 2)])
def test_device_canvas_equals_the_numpy_compositor(pkg, dtype, nx, ny, field, scale):
    """The page's canvas composited per pixel on the GPU (csrc/canvas.hpp: wt_canvas_stroke, wt_canvas_compose) against the NumPy compositor
    of round 3 (compose.py), which draws by the same rules in the same arithmetic: field resampling, particle layer after 25 frames of
    fading strokes, foil fill and outline, bar, labels.  The two differ by at most one level in a handful of pixels (libm vs device hypot
    at rounding ties)."""
    from airfoil_cfd_tool_amd.compose import DeviceTrailLayer, TrailLayer
    from airfoil_cfd_tool_amd.tracers import Tracers
    with pkg.WindTunnel(shape="naca4412", nx=nx, ny=ny, aoa_deg=9.0, dtype=dtype, field=field) as wt:
        host_layer, dev_layer = TrailLayer(scale), wt.trail_layer(scale)
        assert isinstance(dev_layer, DeviceTrailLayer)
        tr_h, tr_d = Tracers(wt, n=700, seed=11), Tracers(wt, n=700, seed=11)        # the same particles twice: one per layer
        for _ in range(25):
            wt.frame(render=False)
            sh, _ = tr_h.draw(host_layer, 16.0)
            sd, _ = tr_d.draw(dev_layer, 16.0)
            assert np.array_equal(sh, sd)
        ref = wt.compose_frame(trails=host_layer, scale=scale)                       # host path (a compose.TrailLayer selects it)
        img = wt.compose_frame(trails=dev_layer, scale=scale)                        # device path
        assert img.shape == ref.shape == (360 * scale, 680 * scale, 4) and img.dtype == np.uint8
        diff = np.abs(img.astype(int) - ref.astype(int))
        assert diff.max() <= 1, (diff.max(), int((diff > 1).sum()))
        assert (diff > 0).mean() < 2e-3, (diff > 0).mean()
        assert (host_layer.a > 0.3).sum() > 200                                       # strokes exist in what was compared
        # without a particle layer, and again after the angle changed (new labels, new polygon)
        assert np.abs(wt.compose_frame(scale=scale).astype(int) - pkg.compose.compose(
            wt.render_rgba()[::-1], wt.geometry.xp, wt.geometry.yp, wt.aoa_deg, pkg.FIELD_MODES[field], wt.y_half_world(), scale=scale).astype(int)).max() <= 1
        wt.aoa_deg = -4.5
        wt.frame(render=False)
        a = wt.compose_frame(trails=dev_layer, scale=scale)
        b = wt.compose_frame(trails=host_layer, scale=scale)
        assert np.abs(a.astype(int) - b.astype(int)).max() <= 1


def test_frame_loop_with_the_device_canvas_is_interactive(pkg):
    """The page's loop (frame(), tracers, composited canvas) on the reference's own lattice: with the canvas on the device it runs far above
    the 9 frames per second the NumPy compositor gave (profiles/r04_a_frame_loop_host_canvas.txt); asserted loosely: > 60 per second."""
    import time
    from airfoil_cfd_tool_amd.tracers import Tracers
    with pkg.WindTunnel(shape="naca2412", nx=320, ny=160, aoa_deg=6.0) as wt:
        layer = wt.trail_layer(1)
        tr = Tracers(wt, seed=5)
        for _ in range(10):
            wt.frame(render=False); tr.draw(layer, 16.0); wt.compose_frame(trails=layer)
        t0 = time.perf_counter()
        for _ in range(60):
            wt.frame(render=False); tr.draw(layer, 16.0); img = wt.compose_frame(trails=layer)
        fps = 60 / (time.perf_counter() - t0)
        assert img.shape == (360, 680, 4) and fps > 60, fps


def test_canvas_at_another_scale_neither_wipes_live_trails_nor_loses_the_labels(pkg):
    """A handle holds one canvas at one scale: composing at another scale while tracer strokes live on the particle layer is REFUSED
    (it used to free the layer silently); once the layer is cleared the canvas may change scale, and the labels are uploaded again for the new
    canvas although the host object remembers having sent them for the old one."""
    from airfoil_cfd_tool_amd import compose
    with pkg.WindTunnel(shape="naca2412", nx=320, ny=160, aoa_deg=6.0) as wt:
        layer, tr = wt.trail_layer(2), pkg.Tracers(wt, n=300, seed=5)
        for _ in range(4):
            wt.frame(render=False)
            tr.draw(layer, 16.0)
        with_trails = wt.compose_frame(trails=layer, scale=2)
        assert wt.engine.get_option("canvas_layer_live") == 1.0 and wt.engine.get_option("canvas_scale") == 2.0
        with pytest.raises(pkg.WTError) as ei:
            wt.compose_frame(scale=1)                                     # would re-allocate the canvas and wipe the scale-2 layer
        assert ei.value.code == -5 and "wipe" in str(ei.value)
        again = wt.compose_frame(trails=layer, scale=2)                   # the trails are still there
        assert np.array_equal(again, with_trails)
        wt.engine.canvas_stroke(2, 2)                                     # clear the layer: the canvas may now change scale
        small = wt.compose_frame(scale=1)
        ref = compose.compose(wt.render_rgba()[::-1], wt.geometry.xp, wt.geometry.yp, wt.aoa_deg, 0, wt.y_half_world())
        assert small.shape == (360, 680, 4) and int(np.abs(small.astype(int) - ref.astype(int)).max()) <= 1      # labels included
        big = wt.compose_frame(scale=2)                                   # back to scale 2: same (scale, field, angle) key as the first frames,
        ref2 = compose.compose(wt.render_rgba()[::-1], wt.geometry.xp, wt.geometry.yp, wt.aoa_deg, 0, wt.y_half_world(), scale=2)
        assert int(np.abs(big.astype(int) - ref2.astype(int)).max()) <= 1  # yet the labels are there: the new canvas had none and got them again
