"""CPU: the host compositor (airfoil-cfd-tool_amd/compose.py) against the drawing rules of the reference page
(pages/airfoil_flow_lbm_aerolab.html:704-719, 780-860, 919-927).  No browser canvas exists here, so the image is pinned
by properties: size, plot rectangle, colour-bar rows equal to the CPU colour maps incl. their first and
last stops, the foil filled and outlined exactly around the panel polygon, tick labels and captions present at their
anchors, tracer strokes where particles moved and fading afterwards."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401  (adds the repo to sys.path)
import airfoil_cfd_tool_amd as pkg
from airfoil_cfd_tool_amd import compose


def _frame(mode=0, scale=1, aoa=6.0, trails=None, shape="naca2412"):
    nx, ny = 320, 160
    g = pkg.geometry.build_geometry(nx, ny, aoa, None, shape)
    field = np.zeros((ny, nx, 4), np.uint8)
    field[..., 0] = 40; field[..., 1] = 90; field[..., 2] = 160; field[..., 3] = 255          # a flat blue field
    img = compose.compose(field, g.xp, g.yp, aoa, mode, pkg.geometry.domain_y_half(nx, ny), trails=trails, scale=scale)
    return img, g


@pytest.mark.parametrize("scale", [1, 2])
def test_size_background_and_plot_rectangle(scale):
    img, _ = _frame(scale=scale)
    assert img.shape == (360 * scale, 680 * scale, 4) and img.dtype == np.uint8 and (img[..., 3] == 255).all()   # html:69
    assert tuple(img[2, 2, :3]) == (0x0a, 0x0d, 0x18)                                                           # html:919
    px, py, pw, ph = 54 * scale, 26 * scale, (680 - 54 - 42) * scale, (360 - 52) * scale                        # html:71-72
    assert tuple(img[py + 3, px + 3, :3]) == (40, 90, 160) and tuple(img[py + ph - 4, px + pw - 4, :3]) == (40, 90, 160)
    assert tuple(img[py - 2, px + 5, :3]) == (0x0a, 0x0d, 0x18) and tuple(img[py + 5, px + pw + 1, :3]) == (0x0a, 0x0d, 0x18)


@pytest.mark.parametrize("mode,top,bottom", [(0, (220, 20, 0), (5, 5, 20)), (1, (205, 50, 25), (20, 50, 160)), (2, (250, 71, 46), (38, 128, 250))])
def test_colour_bar_rows_follow_the_cpu_colour_maps(mode, top, bottom):
    """drawBar (html:830-839): row i = map(1 - i/bh) (vorticity: 1 - 2i/bh), channels truncated; first row = the map's
    last stop, the bottom end approaches its first stop."""
    img, _ = _frame(mode)
    bx, by, bw, bh = 680 - 32, 26, 10, 308
    assert tuple(img[by, bx + 4, :3]) == top
    i = np.arange(bh)
    if mode == 0:
        ref = compose.lerp_scale(1 - i / bh, compose.SPEED_SCALE)
    elif mode == 1:
        ref = compose.lerp_scale(1 - i / bh, compose.CP_SCALE)
    else:
        ref = compose.cmap_vort(1 - 2 * i / bh)
    assert np.array_equal(img[by:by + bh, bx + 4, :3], ref.astype(int))
    assert np.abs(img[by + bh - 1, bx + 4, :3].astype(int) - np.array(bottom)).max() <= 6
    assert (img[by:by + bh, bx:bx + bw, :3] == img[by:by + bh, bx:bx + 1, :3]).all()                           # bw = 10 columns alike
    assert tuple(img[by + 40, bx - 2, :3]) == (0x0a, 0x0d, 0x18) and tuple(img[by + 40, bx + bw + 1, :3]) == (0x0a, 0x0d, 0x18)


def test_colour_maps_match_reference_stops():
    """cmap / cmapCp / cmapVort (html:704-719) at their knots."""
    for k, stop in enumerate(compose.SPEED_SCALE):
        assert np.allclose(compose.cmap(k / 9), stop)
    for k, stop in enumerate(compose.CP_SCALE):
        assert np.allclose(compose.cmap_cp(k / 7), stop)
    assert np.allclose(compose.cmap_vort(-1.0), [38, 128, 250]) and np.allclose(compose.cmap_vort(1.0), [250, 71, 46])
    assert np.allclose(compose.cmap_vort(0.0), [15, 18, 28]) and np.allclose(compose.cmap(-3.0), [5, 5, 20]) and np.allclose(compose.cmap(7.0), [220, 20, 0])


@pytest.mark.parametrize("scale,aoa,shape", [(1, 6.0, "naca2412"), (2, -12.0, "naca4412"), (3, 20.0, "naca0012")])
def test_foil_is_filled_and_outlined_on_the_polygon(scale, aoa, shape):
    """drawFoil (html:815-828): interior '#0d1018', a light outline ON the panel polygon, nothing light elsewhere."""
    img, g = _frame(0, scale, aoa, shape=shape)
    cv = compose.Canvas(scale)
    cx, cy = cv.w2c(g.xp, g.yp, pkg.geometry.domain_y_half(320, 160))
    rgb = img[..., :3].astype(int)
    light = (rgb[..., 0] > 120) & (rgb[..., 1] > 130) & (rgb[..., 2] > 170) & (rgb[..., 1] > rgb[..., 0])
    plot = np.zeros_like(light); plot[cv.py:cv.py + cv.ph, cv.px:cv.px + cv.pw] = True
    plot[cv.py:cv.py + 22 * scale, cv.px:cv.px + 130 * scale] = False           # the angle read-out (html:856-859)
    ys, xs = np.nonzero(light & plot)
    assert len(ys) > 100 * scale
    # every outline pixel lies within (width/2 + 1) pixels of the polygon ...
    x2, y2 = np.roll(cx, -1), np.roll(cy, -1)
    d = np.full(len(ys), np.inf)
    for a, b, c, e in zip(cx, cy, x2, y2):
        dx, dy = c - a, e - b
        t = np.clip(((xs + 0.5 - a) * dx + (ys + 0.5 - b) * dy) / max(dx * dx + dy * dy, 1e-12), 0, 1)
        d = np.minimum(d, np.hypot(xs + 0.5 - (a + t * dx), ys + 0.5 - (b + t * dy)))
    assert d.max() <= 0.7 * scale + 1.0
    # ... and every polygon vertex has an outline pixel next to it
    for a, b in zip(cx[::8], cy[::8]):
        assert light[int(b) - 1:int(b) + 2, int(a) - 1:int(a) + 2].any()
    # the interior is the foil colour: a point well inside the section (quarter chord, mid thickness)
    k = int(np.argmin(np.abs(np.asarray(g.xp) - 0.3)))
    inside_x, _ = cv.w2c(0.3, 0.0, pkg.geometry.domain_y_half(320, 160))
    col = rgb[:, int(inside_x)]
    dark = (col == np.array([0x0d, 0x10, 0x18])).all(axis=1)
    assert dark.sum() >= 4 * scale and k >= 0


def test_labels_and_captions_are_drawn_at_their_anchors():
    """drawLabels / drawBar captions (html:840-860): x ticks under the plot at x = 0, 0.5, 1; y ticks left of it at
    -0.4, 0, 0.4; the angle read-out in the top-left corner; 'fast' / 'slow' right of the bar (CCW/CW, +Cp/-Cp)."""
    img, _ = _frame(0)
    cv = compose.Canvas(1)
    bg = np.array([0x0a, 0x0d, 0x18])
    ink = (np.abs(img[..., :3].astype(int) - bg).sum(axis=2) > 60)
    yh = pkg.geometry.domain_y_half(320, 160)
    for xv in (0.0, 0.5, 1.0):
        x, _ = cv.w2c(xv, 0.0, yh)
        assert ink[360 - 8 - 8:360 - 7, int(x) - 10:int(x) + 10].any()
    for yv in (-0.4, 0.0, 0.4):
        _, y = cv.w2c(0.0, yv, yh)
        assert ink[int(y) - 6:int(y) + 5, 54 - 6 - 26:54 - 5].any()
    assert not ink[100:110, 2:20].any()                                     # margins stay clean away from the labels
    bx = 680 - 32 + 10 + 3
    assert ink[26:26 + 10, bx:bx + 24].any() and ink[26 + 308 - 9:26 + 308, bx:bx + 24].any()
    img2, _ = _frame(0, aoa=-7.5)
    a, b = img[26 + 6:26 + 18, 54 + 8:54 + 110, :3], img2[26 + 6:26 + 18, 54 + 8:54 + 110, :3]
    assert (a != b).any()                                                   # the read-out shows the angle


def test_tracer_strokes_fade_and_follow_the_segments():
    """stepParticles' drawing (html:781-803): a stroke where a particle moved, tinted by the speed map, fading by 5.5 %
    of its remaining opacity per frame."""
    layer = compose.TrailLayer(1)
    cv = compose.Canvas(1)
    yh = pkg.geometry.domain_y_half(320, 160)
    seg = np.array([[0.2, 0.30, 0.26, 0.31], [0.9, -0.35, 0.97, -0.35]])
    layer.fade(); layer.stroke(cv, seg, np.array([1.0, 0.0]), yh)
    x0, y0 = cv.w2c(0.23, 0.305, yh)
    a0 = layer.a[int(y0), int(x0)]
    assert 0.5 < a0 <= 0.75 + 1e-9 and layer.a[5, 5] == 0
    fast = layer.rgb[int(y0), int(x0)] / a0
    assert np.abs(fast - np.rint(np.array([220, 20, 0]) * 0.4 + 255 * 0.6 * 1.0)).max() <= 2          # html:796-798, t = 1
    x1, y1 = cv.w2c(0.93, -0.35, yh)
    slow = layer.rgb[int(y1), int(x1)] / layer.a[int(y1), int(x1)]
    assert np.abs(slow - np.rint(np.array([5, 5, 20]) * 0.4 + 255 * 0.6 * 0.55)).max() <= 2           # t = 0: lum 0.55
    for _ in range(10):
        layer.fade()
    assert abs(layer.a[int(y0), int(x0)] / a0 - (1 - 0.055) ** 10) < 1e-9
    img_with, _ = _frame(0, trails=layer)
    img_without, _ = _frame(0)
    assert (img_with[int(y0), int(x0), :3] != img_without[int(y0), int(x0), :3]).any()
    assert (img_with[200, 300] == img_without[200, 300]).all()


def test_png_roundtrip(tmp_path):
    import struct
    import zlib
    from airfoil_cfd_tool_amd.windtunnel import write_png
    img, _ = _frame(1)
    p = tmp_path / "x.png"
    write_png(str(p), img)
    raw = p.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    w, h = struct.unpack(">II", raw[16:24])
    assert (w, h) == (680, 360)
    i = raw.index(b"IDAT")
    n = struct.unpack(">I", raw[i - 4:i])[0]
    data = zlib.decompress(raw[i + 4:i + 4 + n])
    rows = np.frombuffer(data, np.uint8).reshape(h, 1 + 4 * w)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(h, w, 4), img)


def test_device_canvas_inputs_describe_the_host_drawing(pkg):
    """What wt_canvas_compose / wt_canvas_stroke are fed (compose.bar_rows, text_alpha_map, stroke_records) is what the NumPy compositor draws."""
    import numpy as np
    from airfoil_cfd_tool_amd import compose
    cv = compose.Canvas(1)
    img = np.zeros((160, 320, 4), np.uint8)
    ref = compose.compose(img, [1.0, 0.5, 0.0, 0.5], [0.0, 0.05, 0.0, -0.05], 6.0, 2, 0.46)
    bar = compose.bar_rows(2, 1)
    assert bar.shape == (308, 3) and bar.dtype == np.uint8
    assert (ref[26:26 + 308, 680 - 32:680 - 22, :3] == bar[:, None, :]).all()          # the bar's pixels are the rows
    txt = compose.text_alpha_map(1, 6.0, 2, 0.46)
    assert txt.shape == (360, 680) and txt.dtype == np.float32 and set(np.unique(txt)) == {np.float32(0), np.float32(0.4), np.float32(0.55), np.float32(0.75)}
    # label pixels over the plain background: BG blended with white by exactly that alpha
    ys, xs = np.nonzero(txt == np.float32(0.75))
    bg = np.array(compose.BG, dtype=np.float64)
    want = np.rint(bg * (1 - 0.75) + 255.0 * 0.75)
    on_bg = [(y, x) for y, x in zip(ys, xs) if not (54 <= x < 54 + 584 and 26 <= y < 26 + 308)]
    assert all((ref[y, x, :3] == want).all() for y, x in on_bg)
    assert (compose.text_alpha_map(2, -3.5, 0, 0.46) > 0).sum() > 4 * (txt > 0).sum() * 0.8   # scale 2: glyph cells four times the area
    seg = np.array([[0.10, 0.00, 0.12, 0.01], [0.50, 0.10, 0.50, 0.10]])
    rec = compose.stroke_records(cv, seg, np.array([0.3, 0.9]), 0.46)
    assert rec.shape == (2, 8) and rec[1, 4] == 2 and rec[0, 4] == np.ceil(np.hypot(rec[0, 2] - rec[0, 0], rec[0, 3] - rec[0, 1]) * 2) + 1
    assert compose.stroke_records(cv, np.zeros((0, 4)), np.zeros(0), 0.46).shape == (0, 8)
    # the host layer strokes exactly these records
    layer = compose.TrailLayer(1)
    layer.stroke(cv, seg, np.array([0.3, 0.9]), 0.46)
    x, y = int(rec[1, 0]), int(rec[1, 1])
    assert layer.a[y, x] > 0.3 and np.allclose(layer.rgb[y, x] / layer.a[y, x], rec[1, 5:8])      # one brush dab: premultiplied colour / alpha = the stroke colour
