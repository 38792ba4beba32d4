"""Host-side geometry of the wind tunnel: airfoil generators, AoA rotation, cosine
re-panelling and the scanline rasterisation of the solid mask.

Mirrors the JS geometry block of the reference component,
``pages/airfoil_flow_lbm_aerolab.html`` (``html:LINE`` below): ``naca4`` html:99-116,
``clarkY`` html:118-121, ``SHAPES`` html:123-129, ``rotate`` html:133-140,
``panelise`` html:142-157 (``NP`` html:131), ``rasterMask`` html:160-182,
``buildGeometry`` html:559-577.  In the reference this code runs on the host
(single JS thread) as well; it is O(NY x 160) and stays on the host here.

Arithmetic is IEEE double in the reference's evaluation order so that masks are
identical cell for cell (tests/test_geometry_golden.py compares against the
reference's own JS run under Node).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

# html:73 — world window of the tunnel.  The reference fixes DY to +-0.46 (a 2:1
# lattice); for other aspect ratios the build keeps square cells by scaling the
# y half-height with NY/NX (SURVEY.md §7 "Grid-size generalisation").
DX0, DX1 = -0.42, 1.42
DY_HALF_REF = 0.46
NP = 160  # html:131

Coords = List[Tuple[float, float]]


def domain_y_half(nx: int, ny: int) -> float:
    """Half-height of the tunnel window: the reference's 0.46 on 2:1 lattices."""
    if nx == 2 * ny:
        return DY_HALF_REF
    return 0.5 * (DX1 - DX0) * ny / nx


def naca4(m: float, p: float, t: float, n: int) -> Coords:
    """html:99-116: NACA 4-digit, closed trailing edge (-0.1036), cosine spacing,
    returned in Selig order TE -> upper -> LE -> lower -> TE (2n+1 points)."""
    m /= 100
    p /= 10
    t /= 100
    up: Coords = []
    lo: Coords = []
    for i in range(n + 1):
        b = math.pi * i / n
        x = 0.5 * (1 - math.cos(b))
        yt = 5 * t * (0.2969 * math.sqrt(x) - 0.126 * x - 0.3516 * x * x + 0.2843 * x ** 3 - 0.1036 * x ** 4)
        yc = 0.0
        dyc = 0.0
        if m > 0:
            if x < p:
                yc = m / p / p * (2 * p * x - x * x)
                dyc = 2 * m / p / p * (p - x)
            else:
                yc = m / (1 - p) ** 2 * ((1 - 2 * p) + 2 * p * x - x * x)
                dyc = 2 * m / (1 - p) ** 2 * (p - x)
        th = math.atan(dyc)
        up.append((x - yt * math.sin(th), yc + yt * math.cos(th)))
        lo.append((x + yt * math.sin(th), yc - yt * math.cos(th)))
    up.reverse()
    return up + lo[1:]


_CLARK_Y = [[100, .44], [95, 1.46], [90, 2.22], [80, 3.69], [70, 5.07], [60, 6.23], [50, 7.1], [40, 7.62],
            [30, 7.79], [25, 7.67], [20, 7.35], [15, 6.79], [10, 5.88], [7.5, 5.23], [5, 4.39], [2.5, 3.18],
            [1.25, 2.17], [0, 0], [1.25, -1.35], [2.5, -1.93], [5, -2.55], [7.5, -2.9], [10, -3.05],
            [15, -3.01], [20, -2.75], [25, -2.41], [30, -2.06], [40, -1.38], [50, -.85], [60, -.44],
            [70, -.16], [80, 0], [90, 0], [95, 0], [100, -.44]]


def clark_y() -> Coords:
    """html:118-121: tabulated Clark-Y ordinates in percent chord."""
    return [(x / 100, y / 100) for x, y in _CLARK_Y]


# html:123-129
SHAPES: Dict[str, Callable[[], Coords]] = {
    "naca0012": lambda: naca4(0, 0, 12, 50),
    "naca2412": lambda: naca4(2, 4, 12, 50),
    "naca4412": lambda: naca4(4, 4, 12, 50),
    "naca6409": lambda: naca4(6, 4, 9, 50),
    "clark_y": clark_y,
}


def rotate(coords: Sequence[Tuple[float, float]], a_deg: float) -> Coords:
    """html:133-140: rotate by -a_deg about the quarter chord (0.25, 0):
    positive AoA = nose up, the free stream stays along +x."""
    a = -a_deg * math.pi / 180
    ca, sa = math.cos(a), math.sin(a)
    px, py = 0.25, 0.0
    out: Coords = []
    for x, y in coords:
        dx, dy = x - px, y - py
        out.append((px + dx * ca - dy * sa, py + dx * sa + dy * ca))
    return out


def panelise(coords: Sequence[Tuple[float, float]]) -> Tuple[np.ndarray, np.ndarray]:
    """html:142-157: arc-length cosine resampling to NP+1 = 161 points."""
    xs = [p[0] for p in coords]
    ys = [p[1] for p in coords]
    arc = [0.0]
    for i in range(1, len(coords)):
        arc.append(arc[i - 1] + math.hypot(xs[i] - xs[i - 1], ys[i] - ys[i - 1]))
    L = arc[-1]
    xp = np.empty(NP + 1)
    yp = np.empty(NP + 1)
    for i in range(NP + 1):
        s = L * 0.5 * (1 - math.cos(math.pi * i / NP))
        j = 0
        while j < len(arc) - 2 and arc[j + 1] < s:
            j += 1
        t = (s - arc[j]) / (arc[j + 1] - arc[j] + 1e-12)
        xp[i] = xs[j] + (xs[j + 1] - xs[j]) * t
        yp[i] = ys[j] + (ys[j + 1] - ys[j]) * t
    return xp, yp


def raster_mask(xp: np.ndarray, yp: np.ndarray, nx: int, ny: int, y_half: Optional[float] = None) -> np.ndarray:
    """html:160-182: even-odd scanline fill of the (open) panel polyline at row
    centres; returns uint8 [NY][NX] with 255 = solid, row 0 = bottom.

    Quirks kept on purpose: no closing segment, an
    unpaired last crossing is dropped, the x test is on the integer cell index
    (ceil/floor of the crossing in cell units), the y test on row centres."""
    if y_half is None:
        y_half = domain_y_half(nx, ny)
    dy0, dy1 = -y_half, y_half
    mask = np.zeros((ny, nx), dtype=np.uint8)
    x1, x2 = xp[:-1], xp[1:]
    y1, y2 = yp[:-1], yp[1:]
    for iy in range(ny):
        wy = dy0 + (iy + 0.5) / ny * (dy1 - dy0)
        cross = (y1 > wy) != (y2 > wy)
        if not cross.any():
            continue
        xs = x1[cross] + (x2[cross] - x1[cross]) * (wy - y1[cross]) / (y2[cross] - y1[cross])
        xs = np.sort(xs, kind="stable")
        for k in range(0, len(xs) - 1, 2):
            ix0 = math.ceil((xs[k] - DX0) / (DX1 - DX0) * nx)
            ix1 = math.floor((xs[k + 1] - DX0) / (DX1 - DX0) * nx)
            ix0 = max(0, ix0)
            ix1 = min(nx - 1, ix1)
            if ix1 >= ix0:
                mask[iy, ix0:ix1 + 1] = 255
    return mask


@dataclass
class Geometry:
    """html:556 `sol`: what buildGeometry returns (panel mid-points/normals are
    unused by the LBM path and omitted)."""
    xp: np.ndarray
    yp: np.ndarray
    mask: np.ndarray      # uint8 [NY][NX], 255 = solid
    a_deg: float


def round_coords(coords) -> Coords:
    """The 6-decimal rounding of build_lbm_component (pages/Airfoil_Analysis.py:34-36),
    applied before the coordinates reach the component."""
    return [(round(float(x), 6), round(float(y), 6)) for x, y in coords]


def build_geometry(nx: int, ny: int, a_deg: float, user_coords=None, shape: str = "naca2412",
                   y_half: Optional[float] = None) -> Geometry:
    """html:559-577: injected user coordinates win over the built-in shape."""
    if user_coords is not None and len(user_coords) > 0:
        base = [(float(x), float(y)) for x, y in user_coords]
    else:
        if shape not in SHAPES:
            raise ValueError(f"unknown shape {shape!r}; choose from {sorted(SHAPES)}")
        base = SHAPES[shape]()
    xp, yp = panelise(rotate(base, a_deg))
    return Geometry(xp=xp, yp=yp, mask=raster_mask(xp, yp, nx, ny, y_half), a_deg=float(a_deg))
