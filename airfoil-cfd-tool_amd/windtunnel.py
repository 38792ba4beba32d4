"""Python host of the MI355X wind tunnel: the page's input surface over libwindtunnel.

Drop-in for the LBM component of 583phoenix-hue/Airfoil-CFD-Tool:

* :func:`build_lbm_component` takes what ``pages/Airfoil_Analysis.py:20-42``'s
  function of the same name takes (``coords_after``, ``airfoil_name``), applies the
  same 6-decimal rounding (AA.py:34-36) and returns a running :class:`WindTunnel`
  instead of injecting the coordinates into ``airfoil_flow_lbm_aerolab.html``.
* :class:`WindTunnel` keeps the component's controls as attributes/methods —
  angle of attack (html:26, 943-947), field selector (html:32-36, 952-954), flow
  speed U0 (html:41, 956-959) — and its runtime entry points under their JS names:
  ``init_sim`` (initSim html:492), ``apply_geometry`` (html:579), ``sim_step``
  (html:510), ``read_macro`` (html:547), ``update_fields_from_macro`` (html:596),
  ``compute_forces`` (html:650), ``render_field`` (html:530), ``frame`` (html:902),
  ``stats`` (updateStatsUI html:862).

All lattice arithmetic happens in HIP kernels behind the C-ABI (``_capi.Engine``);
this module holds only the host logic the reference also runs on the host.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import geometry as geo
from ._capi import Engine, WT_FIELD_CP, WT_FIELD_SPEED, WT_FIELD_VORT

# html:78, 80, 472, 528
TAU_DEFAULT = 0.58
STEPS_PER_FRAME = 4
U0_DEFAULT = 0.06
VORT_SCALE = 0.06

FIELD_MODES = {"speed": WT_FIELD_SPEED, "cp": WT_FIELD_CP, "vort": WT_FIELD_VORT}   # html:953


def chord_cells(nx: int) -> float:
    """html:77 CHORD_L = NX/(DX1-DX0): lattice cells per unit chord."""
    return nx / (geo.DX1 - geo.DX0)


def tau_from_reynolds(re: float, u0: float, nx: int) -> float:
    """Invert html:865 Re = U0*CHORD_L/NU_L with NU_L=(tau-0.5)/3 (html:79):
    tau = 0.5 + 3*U0*CHORD_L/Re.  (The reference fixes tau=0.58 and only displays Re.)"""
    if not re > 0:
        raise ValueError("Reynolds number must be positive")
    return 0.5 + 3.0 * u0 * chord_cells(nx) / re


def reynolds(u0: float, nx: int, tau: float) -> float:
    """html:865."""
    return u0 * chord_cells(nx) / ((tau - 0.5) / 3.0)


def stall_label(sep_frac: float) -> str:
    """html:869-884: Attached (<5 %), 'x% sep' (<25 %), else 'STALL ≈ x% sep'."""
    pct = int(math.floor(sep_frac * 100 + 0.5))   # Math.round
    if pct < 5:
        return "Attached"
    if pct < 25:
        return f"{pct}% sep"
    return f"STALL ≈ {pct}% sep"


@dataclass
class Stats:
    """What updateStatsUI shows (html:862-885)."""
    cl: Optional[float]
    cd: Optional[float]          # displayed as max(CD, 0) (html:864)
    reynolds: float
    sep_frac: float
    separation: str


class WindTunnel:
    """D2Q9 lattice-Boltzmann wind tunnel around one airfoil on one MI355X."""

    def __init__(self, coords: Optional[Sequence[Sequence[float]]] = None, name: str = "", *,
                 shape: str = "naca2412", nx: int = 320, ny: int = 160, dtype="float32",
                 aoa_deg: float = 6.0, u0: float = U0_DEFAULT, tau: Optional[float] = None,
                 re: Optional[float] = None, field: str = "speed", device: int = 0,
                 y_half: Optional[float] = None):
        if tau is not None and re is not None:
            raise ValueError("give tau or re, not both")
        if field not in FIELD_MODES:
            raise ValueError(f"field must be one of {sorted(FIELD_MODES)}")
        self.nx, self.ny = int(nx), int(ny)
        self.name = name or "Uploaded airfoil"                 # AA.py:37
        self.user_coords = geo.round_coords(coords) if coords is not None and len(coords) else []
        self.shape = shape
        self.u0 = float(u0)
        self.tau = float(tau) if tau is not None else (tau_from_reynolds(re, self.u0, self.nx) if re is not None else TAU_DEFAULT)
        self.field = field
        self.y_half = y_half
        self.engine = self._make_engine(dtype, device)
        self.dtype = self.engine.dtype
        # html:593 initial ranges, html:641 force state, html:594 frame counter
        self.max_s, self.cp_min, self.cp_max = 0.6, -1.0, 1.0
        self.cl_smooth: Optional[float] = None
        self.cd_smooth: Optional[float] = None
        self.sep_frac = 0.0
        self.stat_counter = 0
        self.steps = 0
        self.geometry: Optional[geo.Geometry] = None
        self.macro = None
        self.parser_fixes: list = []
        self.init_sim(self.u0)                                   # html:502-504
        self.apply_geometry(aoa_deg)                             # html:969-970

    # ---- hooks the sharded subclass (distributed.SlabWindTunnel) overrides ----------------
    def _make_engine(self, dtype, device):
        return Engine(self.nx, self.ny, dtype=dtype, device=device)

    def _reduce_ranges(self):
        return self.engine.reduce_ranges(self.u0)

    def _forces(self):
        return self.engine.forces()

    @classmethod
    def from_dat(cls, dat_path: str, name: Optional[str] = None, **kwargs) -> "WindTunnel":
        """The page's `.dat` upload: parse/repair the file like the back end (main.py:543-608 ->
        datfile.load_dat) and feed `coords_after` to the tunnel like AA.py:1413-1416."""
        import os
        from .datfile import load_dat
        coords, fixes = load_dat(dat_path)
        wt = cls(coords, name or os.path.splitext(os.path.basename(dat_path))[0], **kwargs)
        wt.parser_fixes = fixes
        return wt

    # ---- the component's runtime, under its JS names -------------------------------
    def init_sim(self, u0: float) -> None:
        """initSim (html:492-500): uniform equilibrium everywhere."""
        self.engine.init_equilibrium(u0)
        self.steps = 0

    def apply_geometry(self, aoa_deg: float, shape: Optional[str] = None) -> None:
        """applyGeometry (html:579-586): new mask, flow state kept."""
        if shape is not None:
            self.shape = shape
        self.geometry = geo.build_geometry(self.nx, self.ny, aoa_deg, self.user_coords, self.shape, self.y_half)
        self.engine.set_mask(self.geometry.mask)

    @property
    def aoa_deg(self) -> float:
        return self.geometry.a_deg

    @aoa_deg.setter
    def aoa_deg(self, value: float) -> None:          # the AoA slider (html:943-947)
        self.apply_geometry(value)

    def set_flow_speed(self, u0: float) -> None:      # the flow-speed slider (html:956-959): uniform only
        self.u0 = float(u0)

    def set_field(self, field: str) -> None:          # the field selector (html:952-954)
        if field not in FIELD_MODES:
            raise ValueError(f"field must be one of {sorted(FIELD_MODES)}")
        self.field = field

    def sim_step(self, n: int = 1) -> None:
        """n x simStep (html:510-525)."""
        self.engine.step(n, self.tau, self.u0)
        self.steps += n

    def read_macro(self):
        """readMacro (html:547-552): (rho, ux, uy), each [NY][NX]."""
        self.macro = self.engine.read_macro()
        return self.macro

    def update_fields_from_macro(self):
        """updateFieldsFromMacro's range scan (html:596-614) incl. 'keep previous' (611-613)."""
        mx, cmin, cmax = self._reduce_ranges()
        if mx > 0:
            self.max_s = mx
        if math.isfinite(cmin):
            self.cp_min = cmin
        if math.isfinite(cmax):
            self.cp_max = cmax
        return self.max_s, self.cp_min, self.cp_max

    def compute_forces(self):
        """computeForces (html:650-700): pressure force on the staircase body and
        separation fraction, with the reference's exponential smoothing."""
        fx, fy, surf, rev = self._forces()
        if surf == 0:                                   # `if(!any) return;`
            return None
        q = 0.5 * self.u0 * self.u0 * chord_cells(self.nx)
        cl_raw, cd_raw = fy / q, fx / q
        self.cl_smooth = cl_raw if self.cl_smooth is None else self.cl_smooth * 0.9 + cl_raw * 0.1
        self.cd_smooth = cd_raw if self.cd_smooth is None else self.cd_smooth * 0.9 + cd_raw * 0.1
        self.sep_frac = self.sep_frac * 0.85 + (rev / surf) * 0.15
        return cl_raw, cd_raw, rev / surf

    def clamp_events(self):
        """How many fluid sites the stability net (html:344-350) holds at a bound: (density, speed); (0, 0) when healthy."""
        return self.engine.clamp_events()

    def render_field(self, max_s: Optional[float] = None, cp_min: Optional[float] = None,
                     cp_max: Optional[float] = None, field: Optional[str] = None) -> np.ndarray:
        """renderField's field math (html:530-545 + 395-420): the colour-map
        argument t, [NY][NX], NaN on the body."""
        mode = FIELD_MODES[field or self.field]
        return self.engine.field(mode, self.u0,
                                 self.max_s if max_s is None else max_s,
                                 self.cp_min if cp_min is None else cp_min,
                                 self.cp_max if cp_max is None else cp_max, VORT_SCALE)

    def render_rgba(self, field: Optional[str] = None) -> np.ndarray:
        """renderField through RENDER_FS's colour maps (html:371-397): RGBA8 [NY][NX][4], row 0 = bottom,
        using the current (previous frame's) ranges like html:909."""
        mode = FIELD_MODES[field or self.field]
        return self.engine.render_rgba(mode, self.u0, self.max_s, self.cp_min, self.cp_max, VORT_SCALE)

    def y_half_world(self) -> float:
        """Half-height of the tunnel window in chord units (html:73 on 2:1 lattices; square cells otherwise)."""
        return self.y_half if self.y_half is not None else geo.domain_y_half(self.nx, self.ny)

    def _canvas_on_device(self, trails) -> bool:
        """The device canvas (wt_canvas_compose) serves whole-lattice handles; a host-side compose.TrailLayer keeps the NumPy compositor."""
        from . import compose
        return (hasattr(self.engine, "canvas_compose") and getattr(self.engine, "nranks", 1) == 1
                and (trails is None or isinstance(trails, compose.DeviceTrailLayer)))

    def trail_layer(self, scale: int = 1):
        """A particle layer for Tracers.draw: on the GPU where the canvas is composited there, else the NumPy one."""
        from . import compose
        return compose.DeviceTrailLayer(self.engine, scale) if self._canvas_on_device(None) else compose.TrailLayer(scale)

    def compose_frame(self, field: Optional[str] = None, trails=None, scale: int = 1) -> np.ndarray:
        """The page's whole canvas for the current state (html:919-927): field image scaled into the plot rectangle,
        tracer strokes (a particle layer, see trail_layer / tracers.Tracers.draw), foil fill + outline, colour bar with captions,
        axis ticks and the angle read-out.  RGBA8 [360*scale][680*scale][4], top row first.  Composited on the GPU per canvas pixel
        (csrc/canvas.hpp) unless `trails` is a host-side compose.TrailLayer or the tunnel is sharded: then by the NumPy compositor."""
        from . import compose
        mode = FIELD_MODES[field or self.field]
        if not self._canvas_on_device(trails):
            return compose.compose(self.render_rgba(field)[::-1], self.geometry.xp, self.geometry.yp, self.aoa_deg, mode,
                                   self.y_half_world(), trails=trails, scale=scale)
        if trails is not None and (trails.s != int(scale) or trails.engine is not self.engine):
            raise ValueError("the particle layer belongs to another canvas (scale or tunnel)")
        cv = compose.Canvas(scale, alloc=False)
        fx, fy = cv.w2c(self.geometry.xp, self.geometry.yp, self.y_half_world())
        key = (int(scale), mode, float(self.aoa_deg), float(self.y_half_world()))
        text = None
        # the labels change with the angle, the field and the scale only — and must be sent again whenever the library's canvas holds no label map:
        # a canvas (re-)allocated at this scale starts without one, whatever this object remembers
        if (key != getattr(self, "_canvas_text_key", None) or int(self.engine.get_option("canvas_scale")) != int(scale)
                or not self.engine.get_option("canvas_text_set")):
            text = compose.text_alpha_map(scale, self.aoa_deg, mode, self.y_half_world())
        img = self.engine.canvas_compose(scale, mode, self.u0, self.max_s, self.cp_min, self.cp_max, VORT_SCALE, np.column_stack([fx, fy]),
                                         compose.bar_rows(mode, scale), text, trails is not None)
        self._canvas_text_key = key
        return img

    def save_png(self, path: Optional[str] = None, field: Optional[str] = None, composite: bool = True, trails=None,
                 scale: int = 1) -> str:
        """The PNG button (html:980-1000): the composited canvas under the page's file name; composite=False writes the
        bare lattice field at lattice resolution instead."""
        path = path or self.png_name()
        if composite:
            write_png(path, self.compose_frame(field, trails=trails, scale=scale))
        else:
            write_png(path, self.render_rgba(field)[::-1])      # PNG rows run top to bottom
        return path

    def frame(self, render: bool = True):
        """One pass of frame() (html:902-930) without the browser-only parts:
        4 steps, field with the PREVIOUS frame's ranges, range update, forces
        every 3rd frame."""
        self.sim_step(STEPS_PER_FRAME)
        t = self.render_field() if render else None
        self.update_fields_from_macro()
        self.stat_counter += 1
        if self.stat_counter % 3 == 0:
            self.compute_forces()
        return t

    def stats(self) -> Stats:
        """updateStatsUI (html:862-885)."""
        return Stats(cl=self.cl_smooth,
                     cd=None if self.cd_smooth is None else max(self.cd_smooth, 0.0),
                     reynolds=reynolds(self.u0, self.nx, self.tau),
                     sep_frac=self.sep_frac,
                     separation=stall_label(self.sep_frac))

    def png_name(self) -> str:
        """html:990-992."""
        import re as _re
        stem = _re.sub(r"\s+", "_", self.name or "airfoil")
        return f"{stem}_alpha{self.aoa_deg:.1f}deg_lbm.png"

    # ---- test / checkpoint access ---------------------------------------------------
    def read_f(self) -> np.ndarray:
        return self.engine.read_f()

    def write_f(self, f: np.ndarray) -> None:
        self.engine.write_f(f)

    def close(self) -> None:
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def write_png(path: str, rgba: np.ndarray) -> None:
    """Minimal RGBA8 PNG encoder (zlib only; the image has no PIL dependency)."""
    import struct
    import zlib
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("rgba must be [H][W][4] uint8")
    h, w, _ = a.shape
    raw = np.empty((h, 1 + 4 * w), dtype=np.uint8)
    raw[:, 0] = 0                                                # filter type 0 on every scanline
    raw[:, 1:] = a.reshape(h, 4 * w)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        fh.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        fh.write(chunk(b"IEND", b""))


def build_lbm_component(coords_after, airfoil_name: str = "", **kwargs) -> WindTunnel:
    """Same inputs as pages/Airfoil_Analysis.py:20 (``coords_after`` as echoed by the
    back end, main.py:607-608, and a display name); returns the running tunnel."""
    return WindTunnel(coords_after, airfoil_name, **kwargs)
