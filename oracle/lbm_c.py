"""ctypes loader for the C restatement of the oracle (oracle/lbm_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of lbm_oracle.c.  Built by
``make -C oracle`` (``__graft_entry__.build()`` does it); the shared object
lives in oracle/_build/ and travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblbm_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise ValueError("dtype must be float32 or float64")


def equilibrium_init(nx, ny, u0, dtype=np.float32):
    f = np.empty((9, ny, nx), dtype=dtype)
    rho = np.empty((ny, nx), dtype=dtype)
    ux = np.empty_like(rho)
    uy = np.empty_like(rho)
    fn = getattr(lib(), "oracle_init_" + _suffix(dtype))
    fn(_ptr(f), _ptr(rho), _ptr(ux), _ptr(uy), ctypes.c_int(nx), ctypes.c_int(ny), ctypes.c_double(u0))
    return f, (rho, ux, uy)


def step(f, solid, tau, u0):
    _, ny, nx = f.shape
    f = np.ascontiguousarray(f)
    solid = np.ascontiguousarray(solid, dtype=np.uint8)
    fo = np.empty_like(f)
    rho = np.empty((ny, nx), dtype=f.dtype)
    ux = np.empty_like(rho)
    uy = np.empty_like(rho)
    fn = getattr(lib(), "oracle_step_" + _suffix(f.dtype))
    fn(_ptr(f), _ptr(fo), _ptr(rho), _ptr(ux), _ptr(uy), _ptr(solid), ctypes.c_int(nx), ctypes.c_int(ny),
       ctypes.c_double(tau), ctypes.c_double(u0))
    return fo, (rho, ux, uy)


def run(solid, steps, tau=0.58, u0=0.06, dtype=np.float32, f=None):
    """Same contract as lbm_numpy.run."""
    solid = np.ascontiguousarray(solid, dtype=np.uint8)
    ny, nx = solid.shape
    if f is None:
        f, macro = equilibrium_init(nx, ny, u0, dtype)
    else:
        f = np.array(f, dtype=dtype, order="C", copy=True)
        macro = None
    if steps == 0:
        return f, macro
    fb = np.empty_like(f)
    rho = np.empty((ny, nx), dtype=f.dtype)
    ux = np.empty_like(rho)
    uy = np.empty_like(rho)
    fn = getattr(lib(), "oracle_run_" + _suffix(f.dtype))
    fn(_ptr(f), _ptr(fb), _ptr(rho), _ptr(ux), _ptr(uy), _ptr(solid), ctypes.c_int(nx), ctypes.c_int(ny),
       ctypes.c_double(tau), ctypes.c_double(u0), ctypes.c_int(steps))
    return f, (rho, ux, uy)
