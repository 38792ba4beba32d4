"""ctypes binding of libwindtunnel.so (include/windtunnel.h).

The product path: there is no CPU fallback.  If the shared library is missing or
cannot be loaded this module raises; if no HIP device is usable ``wt_create``
fails and :class:`WTError` is raised.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libwindtunnel.so")

WT_OK = 0
WT_F32, WT_F64 = 0, 1
WT_FIELD_SPEED, WT_FIELD_CP, WT_FIELD_VORT = 0, 1, 2
WT_COMM_ID_BYTES = 128

EXPORTS = (
    "wt_create", "wt_create_slab", "wt_create_slab_at", "wt_destroy", "wt_get_info", "wt_last_error", "wt_version",
    "wt_set_option", "wt_get_option",
    "wt_comm_unique_id", "wt_comm_init_rank", "wt_comm_selftest", "wt_link_local", "wt_step_group", "wt_step_group_timed",
    "wt_set_mask", "wt_init_equilibrium", "wt_step", "wt_step_timed", "wt_plan_steps", "wt_read_f", "wt_write_f",
    "wt_read_macro", "wt_reduce_ranges", "wt_forces", "wt_clamp_events", "wt_field", "wt_render_rgba", "wt_advect_tracers",
    "wt_canvas_stroke", "wt_canvas_compose", "wt_sync",
)


class WTError(RuntimeError):
    """A libwindtunnel call returned a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libwindtunnel error {code}: {message}")
        self.code = code


class WtInfo(ctypes.Structure):
    _fields_ = [("nx_global", c_int32), ("ny", c_int32), ("dtype", c_int32), ("device", c_int32),
                ("rank", c_int32), ("nranks", c_int32), ("x0", c_int32), ("width", c_int32),
                ("halo", c_int32), ("reserved", c_int32), ("steps_done", c_int64), ("device_bytes", c_int64)]


_lib = None


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load libwindtunnel.so.  PyTorch ships its own HIP runtime and RCCL; when
    torch is importable it is imported FIRST so that this library binds to the
    same libamdhip64/librccl instances (two HIP runtimes in one process do not
    work)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `make lib` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the wind-tunnel kernels.")
    try:
        import torch  # noqa: F401  (side effect: loads torch's libamdhip64 / librccl first)
    except Exception:  # pragma: no cover - torch is optional for single-GPU use
        pass
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    H = c_void_p
    sig = {
        "wt_create": ([c_int, c_int, c_int, c_int, POINTER(H)], c_int),
        "wt_create_slab": ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(H)], c_int),
        "wt_create_slab_at": ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(H)], c_int),
        "wt_destroy": ([H], c_int),
        "wt_get_info": ([H, POINTER(WtInfo)], c_int),
        "wt_last_error": ([], c_char_p),
        "wt_version": ([], c_char_p),
        "wt_set_option": ([H, c_char_p, c_double], c_int),
        "wt_get_option": ([H, c_char_p, POINTER(c_double)], c_int),
        "wt_comm_unique_id": ([c_void_p], c_int),
        "wt_comm_init_rank": ([H, c_void_p], c_int),
        "wt_comm_selftest": ([c_int, c_int], c_int),
        "wt_link_local": ([POINTER(H), c_int], c_int),
        "wt_step_group": ([POINTER(H), c_int, c_int, c_double, c_double], c_int),
        "wt_step_group_timed": ([POINTER(H), c_int, c_int, c_double, c_double, POINTER(c_float)], c_int),
        "wt_set_mask": ([H, c_void_p], c_int),
        "wt_init_equilibrium": ([H, c_double], c_int),
        "wt_step": ([H, c_int, c_double, c_double], c_int),
        "wt_step_timed": ([H, c_int, c_double, c_double, POINTER(c_float)], c_int),
        "wt_plan_steps": ([H, c_int, c_double, POINTER(c_int), c_int], c_int),
        "wt_read_f": ([H, c_void_p], c_int),
        "wt_write_f": ([H, c_void_p], c_int),
        "wt_read_macro": ([H, c_void_p, c_void_p, c_void_p], c_int),
        "wt_reduce_ranges": ([H, c_double, POINTER(c_double), POINTER(c_double), POINTER(c_double)], c_int),
        "wt_forces": ([H, POINTER(c_double), POINTER(c_double), POINTER(c_int64), POINTER(c_int64)], c_int),
        "wt_clamp_events": ([H, POINTER(c_int64), POINTER(c_int64)], c_int),
        "wt_field": ([H, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p], c_int),
        "wt_render_rgba": ([H, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p], c_int),
        "wt_advect_tracers": ([H, c_int, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_double, c_double,
                               c_void_p, c_void_p, c_void_p, c_void_p], c_int),
        "wt_canvas_stroke": ([H, c_int, c_int, c_int, c_void_p], c_int),
        "wt_canvas_compose": ([H, c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p, c_int, c_void_p, c_void_p, c_int,
                               c_void_p], c_int),
        "wt_sync": ([H], c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != WT_OK:
        raise WTError(rc, load_library().wt_last_error().decode("utf-8", "replace"))


def _np_dtype(dtype) -> np.dtype:
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("dtype must be float32 or float64")
    return dt


class Engine:
    """One libwindtunnel handle (a whole lattice, or one column slab of it)."""

    def __init__(self, nx: int, ny: int, dtype="float32", device: int = 0,
                 rank: int = 0, nranks: int = 1, halo: int = 0, edges=None):
        """edges: the split of a slab tunnel, nranks + 1 rising column indices from 0 to nx (None: equal widths)."""
        self._lib = load_library()
        self.dtype = _np_dtype(dtype)
        self._h = c_void_p()
        code = WT_F32 if self.dtype == np.float32 else WT_F64
        if nranks == 1:
            _check(self._lib.wt_create(nx, ny, code, device, byref(self._h)))
        elif edges is None:
            _check(self._lib.wt_create_slab(nx, ny, code, device, rank, nranks, halo, byref(self._h)))
        else:
            if len(edges) != nranks + 1:
                raise ValueError(f"edges must hold nranks + 1 = {nranks + 1} column indices")
            arr = (c_int * (nranks + 1))(*[int(e) for e in edges])
            _check(self._lib.wt_create_slab_at(nx, ny, code, device, rank, nranks, halo, arr, byref(self._h)))
        info = self.info()
        self.nx_global, self.ny = info.nx_global, info.ny
        self.x0, self.width = info.x0, info.width
        self.rank, self.nranks, self.halo = info.rank, info.nranks, info.halo

    # -- life cycle --
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.wt_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self) -> WtInfo:
        info = WtInfo()
        _check(self._lib.wt_get_info(self._h, byref(info)))
        return info

    def set_option(self, name: str, value: float) -> None:
        _check(self._lib.wt_set_option(self._h, name.encode(), float(value)))

    def get_option(self, name: str) -> float:
        v = c_double()
        _check(self._lib.wt_get_option(self._h, name.encode(), byref(v)))
        return v.value

    # -- transports --
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(WT_COMM_ID_BYTES)
        _check(load_library().wt_comm_unique_id(buf))
        return buf.raw

    def comm_init_rank(self, comm_id: bytes) -> None:
        if len(comm_id) != WT_COMM_ID_BYTES:
            raise ValueError("comm id must be WT_COMM_ID_BYTES long")
        buf = ctypes.create_string_buffer(comm_id, WT_COMM_ID_BYTES)
        _check(self._lib.wt_comm_init_rank(self._h, buf))

    @staticmethod
    def comm_selftest(device: int = 0, ny: int = 4096) -> None:
        _check(load_library().wt_comm_selftest(int(device), int(ny)))

    @staticmethod
    def link_local(engines) -> None:
        arr = (c_void_p * len(engines))(*[e._h for e in engines])
        _check(load_library().wt_link_local(arr, len(engines)))

    @staticmethod
    def step_group(engines, nsteps: int, tau: float, u0: float) -> None:
        arr = (c_void_p * len(engines))(*[e._h for e in engines])
        _check(load_library().wt_step_group(arr, len(engines), int(nsteps), float(tau), float(u0)))

    @staticmethod
    def step_group_timed(engines, nsteps: int, tau: float, u0: float):
        """step_group with each slab's device time (ms) from HIP events on its compute stream."""
        arr = (c_void_p * len(engines))(*[e._h for e in engines])
        ms = (c_float * len(engines))()
        _check(load_library().wt_step_group_timed(arr, len(engines), int(nsteps), float(tau), float(u0), ms))
        return [float(v) for v in ms]

    # -- state --
    def set_mask(self, mask: np.ndarray) -> None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape != (self.ny, self.nx_global):
            raise ValueError(f"mask must have shape [NY][NX] = {(self.ny, self.nx_global)}, got {m.shape}")
        _check(self._lib.wt_set_mask(self._h, m.ctypes.data_as(c_void_p)))

    def init_equilibrium(self, u0: float) -> None:
        _check(self._lib.wt_init_equilibrium(self._h, float(u0)))

    def step(self, nsteps: int, tau: float, u0: float) -> None:
        _check(self._lib.wt_step(self._h, int(nsteps), float(tau), float(u0)))

    def plan_steps(self, nsteps: int, tau: float):
        """The sequence of passes / single steps / refresh steps `step(nsteps)` would take now (+k, 1, -1), without taking them."""
        cap = int(nsteps) + 1
        seq = (c_int * cap)()
        n = self._lib.wt_plan_steps(self._h, int(nsteps), float(tau), seq, cap)
        if n < 0:
            _check(n)
        return [int(v) for v in seq[:n]]

    def step_timed(self, nsteps: int, tau: float, u0: float) -> float:
        ms = c_float()
        _check(self._lib.wt_step_timed(self._h, int(nsteps), float(tau), float(u0), byref(ms)))
        return float(ms.value)

    def read_f(self) -> np.ndarray:
        f = np.empty((9, self.ny, self.width), dtype=self.dtype)
        _check(self._lib.wt_read_f(self._h, f.ctypes.data_as(c_void_p)))
        return f

    def write_f(self, f: np.ndarray) -> None:
        a = np.ascontiguousarray(f, dtype=self.dtype)
        if a.shape != (9, self.ny, self.width):
            raise ValueError(f"f must have shape {(9, self.ny, self.width)}, got {a.shape}")
        _check(self._lib.wt_write_f(self._h, a.ctypes.data_as(c_void_p)))

    # -- read-backs --
    def read_macro(self):
        shape = (self.ny, self.width)
        rho, ux, uy = (np.empty(shape, dtype=self.dtype) for _ in range(3))
        _check(self._lib.wt_read_macro(self._h, rho.ctypes.data_as(c_void_p), ux.ctypes.data_as(c_void_p),
                                       uy.ctypes.data_as(c_void_p)))
        return rho, ux, uy

    def reduce_ranges(self, u0: float):
        a, b, c = c_double(), c_double(), c_double()
        _check(self._lib.wt_reduce_ranges(self._h, float(u0), byref(a), byref(b), byref(c)))
        return a.value, b.value, c.value

    def forces(self):
        fx, fy, surf, rev = c_double(), c_double(), c_int64(), c_int64()
        _check(self._lib.wt_forces(self._h, byref(fx), byref(fy), byref(surf), byref(rev)))
        return fx.value, fy.value, surf.value, rev.value

    def clamp_events(self):
        """(sites at a density bound, sites at the speed bound) of the last emitted state (html:344-350)."""
        a, b = c_int64(), c_int64()
        _check(self._lib.wt_clamp_events(self._h, byref(a), byref(b)))
        return a.value, b.value

    def field(self, mode: int, u0: float, max_s: float, cp_min: float, cp_max: float, vort_scale: float) -> np.ndarray:
        out = np.empty((self.ny, self.width), dtype=self.dtype)
        _check(self._lib.wt_field(self._h, int(mode), float(u0), float(max_s), float(cp_min), float(cp_max),
                                  float(vort_scale), out.ctypes.data_as(c_void_p)))
        return out

    def render_rgba(self, mode: int, u0: float, max_s: float, cp_min: float, cp_max: float, vort_scale: float) -> np.ndarray:
        out = np.empty((self.ny, self.width, 4), dtype=np.uint8)
        _check(self._lib.wt_render_rgba(self._h, int(mode), float(u0), float(max_s), float(cp_min), float(cp_max),
                                        float(vort_scale), out.ctypes.data_as(c_void_p)))
        return out

    def advect_tracers(self, x: np.ndarray, y: np.ndarray, dt_frame: float, u0: float, window):
        """advect() for arrays of particles; returns (x_new, y_new, speed, ok)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if x.shape != y.shape or x.ndim != 1:
            raise ValueError("x and y must be 1-D arrays of equal length")
        n = x.size
        xn, yn, sp = np.empty(n), np.empty(n), np.empty(n)
        ok = np.empty(n, dtype=np.uint8)
        dx0, dx1, dy0, dy1 = (float(v) for v in window)
        _check(self._lib.wt_advect_tracers(self._h, n, x.ctypes.data_as(c_void_p), y.ctypes.data_as(c_void_p), float(dt_frame),
                                           float(u0), dx0, dx1, dy0, dy1, xn.ctypes.data_as(c_void_p),
                                           yn.ctypes.data_as(c_void_p), sp.ctypes.data_as(c_void_p), ok.ctypes.data_as(c_void_p)))
        return xn, yn, sp, ok.astype(bool)

    # -- the page's canvas on the device (csrc/canvas.hpp) --
    def canvas_stroke(self, scale: int, fade: int, seg: Optional[np.ndarray] = None) -> None:
        """Particle layer: fade (0 none / 1 fade / 2 clear), then stroke seg[n][8] = x0, y0, x1, y1 (canvas px), samples, r, g, b in order."""
        if seg is None or len(seg) == 0:
            _check(self._lib.wt_canvas_stroke(self._h, int(scale), int(fade), 0, None))
            return
        a = np.ascontiguousarray(seg, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != 8:
            raise ValueError("seg must be [n][8]")
        _check(self._lib.wt_canvas_stroke(self._h, int(scale), int(fade), int(a.shape[0]), a.ctypes.data_as(c_void_p)))

    def canvas_compose(self, scale: int, mode: int, u0: float, max_s: float, cp_min: float, cp_max: float, vort_scale: float,
                       poly_xy: np.ndarray, bar_rgb: np.ndarray, text_alpha: Optional[np.ndarray], use_trails: bool) -> np.ndarray:
        """One frame of the page's canvas, RGBA8 [360*scale][680*scale][4], top row first; text_alpha None = the previous call's map."""
        s = int(scale)
        poly = np.ascontiguousarray(poly_xy, dtype=np.float64).reshape(-1, 2)
        bar = np.ascontiguousarray(bar_rgb, dtype=np.uint8)
        if bar.shape != (308 * s, 3):
            raise ValueError(f"bar_rgb must be [{308 * s}][3]")
        txt = None
        if text_alpha is not None:
            txt = np.ascontiguousarray(text_alpha, dtype=np.float32)
            if txt.shape != (360 * s, 680 * s):
                raise ValueError(f"text_alpha must be [{360 * s}][{680 * s}]")
        out = np.empty((360 * s, 680 * s, 4), dtype=np.uint8)
        _check(self._lib.wt_canvas_compose(self._h, s, int(mode), float(u0), float(max_s), float(cp_min), float(cp_max), float(vort_scale),
                                           poly.ctypes.data_as(c_void_p), int(poly.shape[0]), bar.ctypes.data_as(c_void_p),
                                           None if txt is None else txt.ctypes.data_as(c_void_p), 1 if use_trails else 0,
                                           out.ctypes.data_as(c_void_p)))
        return out

    def sync(self) -> None:
        _check(self._lib.wt_sync(self._h))
