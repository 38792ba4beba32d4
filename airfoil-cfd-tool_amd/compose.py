"""The page's 2-D canvas, composited on the host: field image, tracer strokes, foil outline, colour bar, labels.

The reference draws every frame onto a 680 x 360 canvas (``pages/airfoil_flow_lbm_aerolab.html``): dark background,
the WebGL field scaled into the plot rectangle (html:919-923), the particle layer (fading strokes, html:780-808), the
filled and outlined foil (html:815-828), the colour bar with its two captions (html:830-848), the axis ticks and the
angle read-out (html:850-860); the PNG button saves exactly that canvas (html:980-1000).  This module reproduces those
drawing rules with NumPy on an RGBA float canvas of ``scale`` x (680 x 360) pixels.  It is presentation code — a
browser's anti-aliasing and font cannot be reproduced bit for bit — so the tests pin it by properties: image size, plot
rectangle, bar colours equal to the CPU colour maps' stops (html:704-719), outline pixels on the polygon's edge, the
interior of the foil filled, strokes where the tracers moved.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import geometry as geo

W0, H0 = 680, 360                      # html:69
PX0, PY0 = 54, 26                      # html:71
PW0, PH0 = W0 - PX0 - 42, H0 - 2 * PY0  # html:72

BG = (0x0a, 0x0d, 0x18)                # '#0a0d18' html:919
FOIL_FILL = (0x0d, 0x10, 0x18)         # '#0d1018' html:824
FOIL_STROKE = (200, 215, 255, 0.85)    # html:826
SPEED_SCALE = [[5, 5, 20], [0, 20, 120], [0, 60, 200], [0, 140, 220], [0, 220, 220], [0, 210, 140], [80, 200, 0], [220, 210, 0], [255, 120, 0], [220, 20, 0]]
CP_SCALE = [[20, 50, 160], [40, 110, 210], [100, 175, 235], [190, 220, 245], [248, 248, 248], [248, 214, 140], [240, 150, 60], [205, 50, 25]]


def lerp_scale(t, scale):
    """lerpScale (html:704-710) for an array of t; returns [..., 3] floats (0..255)."""
    s = np.asarray(scale, dtype=np.float64)
    t = np.clip(np.asarray(t, dtype=np.float64), 0.0, 1.0)
    f = t * (len(s) - 1)
    i = np.minimum(np.floor(f).astype(int), len(s) - 2)
    u = (f - i)[..., None]
    return s[i] * (1 - u) + s[i + 1] * u


def cmap(t):
    return lerp_scale(t, SPEED_SCALE)                       # html:713


def cmap_cp(t):
    return lerp_scale(t, CP_SCALE)                          # html:714


def cmap_vort(t):
    """html:715-719."""
    t = np.clip(np.asarray(t, dtype=np.float64), -1.0, 1.0)
    neg = lerp_scale(-t, [[15, 18, 28], [38, 128, 250]])
    pos = lerp_scale(t, [[15, 18, 28], [250, 71, 46]])
    return np.where((t < 0)[..., None], neg, pos)


# 5 x 7 bitmap glyphs for the handful of characters the page writes (ticks, bar captions, the angle read-out)
_GLYPHS = {
    "0": "01110 10001 10011 10101 11001 10001 01110", "1": "00100 01100 00100 00100 00100 00100 01110",
    "2": "01110 10001 00001 00010 00100 01000 11111", "3": "11110 00001 00001 01110 00001 00001 11110",
    "4": "00010 00110 01010 10010 11111 00010 00010", "5": "11111 10000 11110 00001 00001 10001 01110",
    "6": "00110 01000 10000 11110 10001 10001 01110", "7": "11111 00001 00010 00100 01000 01000 01000",
    "8": "01110 10001 10001 01110 10001 10001 01110", "9": "01110 10001 10001 01111 00001 00010 01100",
    ".": "00000 00000 00000 00000 00000 01100 01100", "-": "00000 00000 00000 11111 00000 00000 00000",
    "+": "00000 00100 00100 11111 00100 00100 00000", "=": "00000 00000 11111 00000 11111 00000 00000",
    " ": "00000 00000 00000 00000 00000 00000 00000", "°": "01100 10010 10010 01100 00000 00000 00000",
    "α": "00000 00000 01101 10010 10010 10010 01101", "a": "00000 00000 01110 00001 01111 10001 01111",
    "f": "00110 01001 01000 11100 01000 01000 01000", "s": "00000 00000 01111 10000 01110 00001 11110",
    "t": "01000 01000 11100 01000 01000 01001 00110", "l": "01100 00100 00100 00100 00100 00100 01110",
    "o": "00000 00000 01110 10001 10001 10001 01110", "w": "00000 00000 10001 10001 10101 10101 01010",
    "p": "00000 00000 11110 10001 11110 10000 10000", "C": "01110 10001 10000 10000 10000 10001 01110",
    "W": "10001 10001 10001 10101 10101 11011 10001",
}


class Canvas:
    """RGB float canvas (0..255) with the few primitives the page uses."""

    def __init__(self, scale: int = 1, alloc: bool = True):
        """alloc=False: geometry only (w2c) — what a stroke onto a device-resident layer needs."""
        self.s = int(scale)
        self.w, self.h = W0 * self.s, H0 * self.s
        self.px, self.py, self.pw, self.ph = PX0 * self.s, PY0 * self.s, PW0 * self.s, PH0 * self.s
        self.rgb = None
        if alloc:
            self.rgb = np.empty((self.h, self.w, 3), dtype=np.float64)
            self.rgb[:] = BG
        self._cov = None

    # world -> canvas (html:810-811); y_half = half-height of the tunnel window
    def w2c(self, x, y, y_half):
        cx = self.px + (np.asarray(x, dtype=np.float64) - geo.DX0) / (geo.DX1 - geo.DX0) * self.pw
        cy = self.py + (1.0 - (np.asarray(y, dtype=np.float64) + y_half) / (2.0 * y_half)) * self.ph
        return cx, cy

    def blend(self, ys, xs, colour, alpha):
        a = np.asarray(alpha, dtype=np.float64)[..., None]
        self.rgb[ys, xs] = self.rgb[ys, xs] * (1 - a) + np.asarray(colour, dtype=np.float64) * a

    def draw_field(self, rgba_top_first: np.ndarray) -> None:
        """drawImage(glcv, PX, PY, PW, PH) (html:923): bilinear resampling of the lattice image into the plot rectangle."""
        img = np.asarray(rgba_top_first, dtype=np.float64)[..., :3]
        ny, nx = img.shape[:2]
        fx = (np.arange(self.pw) + 0.5) / self.pw * nx - 0.5
        fy = (np.arange(self.ph) + 0.5) / self.ph * ny - 0.5
        x0 = np.clip(np.floor(fx).astype(int), 0, nx - 1); x1 = np.clip(x0 + 1, 0, nx - 1); tx = np.clip(fx - x0, 0, 1)[None, :, None]
        y0 = np.clip(np.floor(fy).astype(int), 0, ny - 1); y1 = np.clip(y0 + 1, 0, ny - 1); ty = np.clip(fy - y0, 0, 1)[:, None, None]
        top = img[y0][:, x0] * (1 - tx) + img[y0][:, x1] * tx
        bot = img[y1][:, x0] * (1 - tx) + img[y1][:, x1] * tx
        self.rgb[self.py:self.py + self.ph, self.px:self.px + self.pw] = top * (1 - ty) + bot * ty

    def polyline(self, cx, cy, colour, alpha: float, width: float, closed: bool = False) -> None:
        """Anti-aliased stroke: coverage = clamp(width/2 + 0.5 - distance to the segment, 0, 1)."""
        cx, cy = np.asarray(cx, dtype=np.float64), np.asarray(cy, dtype=np.float64)
        n = len(cx)
        segs = [(i, (i + 1) % n) for i in range(n if closed else n - 1)]
        r = width / 2.0 + 0.5
        for i, j in segs:
            x0, y0, x1, y1 = cx[i], cy[i], cx[j], cy[j]
            xa, xb = int(np.floor(min(x0, x1) - r)), int(np.ceil(max(x0, x1) + r))
            ya, yb = int(np.floor(min(y0, y1) - r)), int(np.ceil(max(y0, y1) + r))
            xa, ya, xb, yb = max(xa, 0), max(ya, 0), min(xb, self.w - 1), min(yb, self.h - 1)
            if xb < xa or yb < ya:
                continue
            yy, xx = np.mgrid[ya:yb + 1, xa:xb + 1]
            px, py = xx + 0.5, yy + 0.5
            dx, dy = x1 - x0, y1 - y0
            L2 = dx * dx + dy * dy
            t = np.clip(((px - x0) * dx + (py - y0) * dy) / L2, 0.0, 1.0) if L2 > 0 else np.zeros_like(px)
            d = np.hypot(px - (x0 + t * dx), py - (y0 + t * dy))
            c = np.clip(r - d, 0.0, 1.0)
            # max-combine the coverage of the segments of one path (a path is stroked once, joints are not darker)
            tile = self._cov_tile(ya, yb, xa, xb)
            np.maximum(tile, c, out=tile)
        self._flush_cov(colour, alpha)

    def _cov_tile(self, ya, yb, xa, xb):
        if self._cov is None:
            self._cov = np.zeros((self.h, self.w), dtype=np.float64)
        return self._cov[ya:yb + 1, xa:xb + 1]

    def _flush_cov(self, colour, alpha):
        if self._cov is None:
            return
        a = (self._cov * alpha)[..., None]
        self.rgb = self.rgb * (1 - a) + np.asarray(colour, dtype=np.float64) * a
        self._cov = None

    def fill_polygon(self, cx, cy, colour) -> None:
        """Even-odd scanline fill at pixel centres (the canvas default for a simple closed path)."""
        cx, cy = np.asarray(cx, dtype=np.float64), np.asarray(cy, dtype=np.float64)
        x2, y2 = np.roll(cx, -1), np.roll(cy, -1)
        ya, yb = max(int(np.floor(cy.min())), 0), min(int(np.ceil(cy.max())), self.h - 1)
        for y in range(ya, yb + 1):
            yc = y + 0.5
            m = ((cy <= yc) & (y2 > yc)) | ((y2 <= yc) & (cy > yc))
            if not m.any():
                continue
            xs = np.sort(cx[m] + (yc - cy[m]) / (y2[m] - cy[m]) * (x2[m] - cx[m]))
            for a, b in zip(xs[0::2], xs[1::2]):
                i0, i1 = max(int(np.ceil(a - 0.5)), 0), min(int(np.floor(b - 0.5)), self.w - 1)
                if i1 >= i0:
                    self.rgb[y, i0:i1 + 1] = colour

    def text(self, s: str, x: float, y_baseline: float, colour, alpha: float, align: str = "left", px: int = 10) -> None:
        """5 x 7 bitmap text; `px` = the CSS font size the page asks for (10px / 12px); glyph cell = px * 0.6 wide."""
        k = max(1, int(round(self.s * px / 10.0)))
        adv = 6 * k
        width = adv * len(s)
        x0 = x - (width if align == "right" else width / 2.0 if align == "center" else 0.0)
        y0 = int(round(y_baseline)) - 7 * k
        xi = int(round(x0))
        for ch in s:
            rows = _GLYPHS.get(ch, _GLYPHS[" "]).split()
            g = np.array([[c == "1" for c in row] for row in rows], dtype=bool)
            g = np.kron(g, np.ones((k, k), dtype=bool))
            ys, xs = np.nonzero(g)
            ys, xs = ys + y0, xs + xi
            ok = (ys >= 0) & (ys < self.h) & (xs >= 0) & (xs < self.w)
            self.blend(ys[ok], xs[ok], colour, np.full(int(ok.sum()), alpha))
            xi += adv

    def to_rgba8(self) -> np.ndarray:
        out = np.empty((self.h, self.w, 4), dtype=np.uint8)
        out[..., :3] = np.clip(np.rint(self.rgb), 0, 255).astype(np.uint8)
        out[..., 3] = 255
        return out


class TrailLayer:
    """The particle canvas `pcv` (html:780-808): every frame fades by destination-out alpha 0.055, then the segments of
    the particles that moved are stroked (width 1.1, round caps, alpha 0.75) in a tint of the speed colour map."""

    def __init__(self, scale: int = 1):
        self.s = int(scale)
        self.w, self.h = W0 * self.s, H0 * self.s
        self.rgb = np.zeros((self.h, self.w, 3), dtype=np.float64)     # premultiplied colour
        self.a = np.zeros((self.h, self.w), dtype=np.float64)

    def fade(self) -> None:
        self.rgb *= 1.0 - 0.055
        self.a *= 1.0 - 0.055

    def stroke(self, canvas: Canvas, seg: np.ndarray, t: np.ndarray, y_half: float) -> None:
        if len(seg) == 0:
            return
        # segments are a few pixels long: sample them densely and splat with a small round brush
        rec = stroke_records(canvas, seg, t, y_half)
        x0, y0, x1, y1, n, col = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4].astype(int), rec[:, 5:8]
        r = 1.1 * self.s / 2.0 + 0.5
        for i in range(len(seg)):
            ts = np.linspace(0.0, 1.0, int(n[i]))
            xs, ys = x0[i] + ts * (x1[i] - x0[i]), y0[i] + ts * (y1[i] - y0[i])
            xa, xb = int(np.floor(xs.min() - r)), int(np.ceil(xs.max() + r))
            ya, yb = int(np.floor(ys.min() - r)), int(np.ceil(ys.max() + r))
            xa, ya, xb, yb = max(xa, 0), max(ya, 0), min(xb, self.w - 1), min(yb, self.h - 1)
            if xb < xa or yb < ya:
                continue
            yy, xx = np.mgrid[ya:yb + 1, xa:xb + 1]
            d = np.min(np.hypot(xx[..., None] + 0.5 - xs, yy[..., None] + 0.5 - ys), axis=-1)
            a = np.clip(r - d, 0.0, 1.0) * 0.75
            self.rgb[ya:yb + 1, xa:xb + 1] = self.rgb[ya:yb + 1, xa:xb + 1] * (1 - a[..., None]) + col[i] * a[..., None]
            self.a[ya:yb + 1, xa:xb + 1] = self.a[ya:yb + 1, xa:xb + 1] * (1 - a) + a

    def draw_onto(self, canvas: Canvas) -> None:
        canvas.rgb = canvas.rgb * (1 - self.a[..., None]) + self.rgb        # drawImage(pcv, 0, 0), html:924


def stroke_records(canvas: Canvas, seg: np.ndarray, t: np.ndarray, y_half: float) -> np.ndarray:
    """The strokes of one frame as TrailLayer.stroke splats them, [n][8] = x0, y0, x1, y1 (canvas pixels), points sampled along the segment,
    r, g, b (html:796-798) — the host layer loops over them, the device layer (wt_canvas_stroke) takes the array."""
    if len(seg) == 0:
        return np.zeros((0, 8))
    x0, y0 = canvas.w2c(seg[:, 0], seg[:, 1], y_half)
    x1, y1 = canvas.w2c(seg[:, 2], seg[:, 3], y_half)
    base = cmap(t)
    lum = (0.55 + np.asarray(t) * 0.45)[:, None]
    col = np.rint(base * 0.4 + 255.0 * 0.6 * lum)                      # html:798
    n = np.maximum(2, np.ceil(np.hypot(x1 - x0, y1 - y0) * 2).astype(int) + 1)
    return np.column_stack([x0, y0, x1, y1, n.astype(np.float64), col])


class DeviceTrailLayer:
    """The particle canvas `pcv` kept on the GPU (wt_canvas_stroke): the interface of TrailLayer that Tracers.draw uses — fade(), stroke() —
    with the blending done per pixel on the device instead of per particle in Python (86 ms -> well under a millisecond per frame)."""

    def __init__(self, engine, scale: int = 1):
        self.engine, self.s = engine, int(scale)
        self._geom = Canvas(self.s, alloc=False)
        self._fade = 0
        engine.canvas_stroke(self.s, 2)                                # clear

    def fade(self) -> None:
        self._fade = 1                                                 # applied by the next stroke, in the same kernel

    def stroke(self, canvas, seg: np.ndarray, t: np.ndarray, y_half: float) -> None:
        self.engine.canvas_stroke(self.s, self._fade, stroke_records(self._geom, seg, t, y_half))
        self._fade = 0


def bar_rows(field_mode: int, scale: int) -> np.ndarray:
    """drawBar's rows (html:830-848): row i = map(1 - i/bh) truncated like `r|0`; uint8 [PH0 * scale][3]."""
    bh = PH0 * int(scale)
    i = np.arange(bh, dtype=np.float64)
    rows = cmap_cp(1 - i / bh) if field_mode == 1 else cmap_vort(1 - 2 * i / bh) if field_mode == 2 else cmap(1 - i / bh)
    return rows.astype(int).astype(np.uint8)


def label_calls(cv: Canvas, aoa_deg: float, field_mode: int, y_half: float):
    """The page's text (html:830-860) as (string, x, baseline, alpha, align, px) — drawn by compose() and by text_alpha_map() alike."""
    bx, by, bw, bh = cv.w - 32 * cv.s, cv.py, 10 * cv.s, cv.ph
    top = "+Cp" if field_mode == 1 else "CCW" if field_mode == 2 else "fast"
    bot = "-Cp" if field_mode == 1 else "CW" if field_mode == 2 else "slow"
    calls = [(top, bx + bw + 3 * cv.s, by + 9 * cv.s, 0.55, "left", 10), (bot, bx + bw + 3 * cv.s, by + bh - 1 * cv.s, 0.55, "left", 10)]
    for xv in (0.0, 0.5, 1.0):
        x, _ = cv.w2c(xv, 0.0, y_half)
        calls.append((f"{xv:.1f}", float(x), cv.h - 8 * cv.s, 0.4, "center", 10))
    for yv in (-0.4, 0.0, 0.4):
        _, y = cv.w2c(0.0, yv, y_half)
        calls.append((f"{yv:.1f}", cv.px - 6 * cv.s, float(y) + 3 * cv.s, 0.4, "right", 10))
    calls.append((f"α = {aoa_deg:.1f}°", cv.px + 8 * cv.s, cv.py + 16 * cv.s, 0.75, "left", 12))
    return calls


def text_alpha_map(scale: int, aoa_deg: float, field_mode: int, y_half: float) -> np.ndarray:
    """Alpha of the (white) labels at every canvas pixel, float32 [H][W] — what wt_canvas_compose blends last.  The strings do not overlap."""
    probe = Canvas(scale, alloc=False)
    probe.rgb = np.zeros((probe.h, probe.w, 3), dtype=np.float64)     # blending white (1, 1, 1) onto 0 leaves exactly the alpha
    for s, x, y, alpha, align, px in label_calls(probe, aoa_deg, field_mode, y_half):
        probe.text(s, x, y, (1.0, 1.0, 1.0), alpha, align, px)
    return probe.rgb[..., 0].astype(np.float32)


def compose(field_rgba_top_first: np.ndarray, xp: Sequence[float], yp: Sequence[float], aoa_deg: float, field_mode: int,
            y_half: float, trails: Optional[TrailLayer] = None, scale: int = 1) -> np.ndarray:
    """One frame of the page's canvas (html:919-927): RGBA8 [360*scale][680*scale][4], top row first."""
    cv = Canvas(scale)
    cv.draw_field(field_rgba_top_first)
    if trails is not None:
        trails.draw_onto(cv)
    # drawFoil (html:815-828)
    fx, fy = cv.w2c(xp, yp, y_half)
    cv.fill_polygon(fx, fy, FOIL_FILL)
    cv.polyline(fx, fy, FOIL_STROKE[:3], FOIL_STROKE[3], 1.4 * cv.s, closed=True)
    # drawBar (html:830-848): bh rows, row i = map(1 - i/bh) truncated like `r|0`; each 1.5-px rect is overdrawn by the next
    bx, by, bw, bh = cv.w - 32 * cv.s, cv.py, 10 * cv.s, cv.ph
    rows = bar_rows(field_mode, cv.s).astype(np.float64)
    cv.rgb[by:by + bh, bx:bx + bw] = rows[:, None, :]
    if by + bh < cv.h:      # the last rect's lower half pixel
        cv.rgb[by + bh, bx:bx + bw] = cv.rgb[by + bh, bx:bx + bw] * 0.5 + rows[-1] * 0.5
    # drawLabels (html:850-860) and the bar's captions
    for s_, x, y, alpha, align, px in label_calls(cv, aoa_deg, field_mode, y_half):
        cv.text(s_, x, y, (255, 255, 255), alpha, align, px)
    return cv.to_rgba8()
