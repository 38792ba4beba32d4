"""CPU ORACLE — test infrastructure only (never imported by the product path).

A straight NumPy transcription of the reference's D2Q9 lattice-Boltzmann wind
tunnel, ``pages/airfoil_flow_lbm_aerolab.html`` of 583phoenix-hue/Airfoil-CFD-Tool
(cited below as ``html:LINE``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.

Arithmetic contract: IEEE binary32 (or binary64 for the
double-precision path), literal left-to-right evaluation of the shader's
expressions, one rounding per operation, no FMA contraction, division by tau.

Pinning: this restatement is checked bit-for-bit against the reference's own
shader text executed headless (oracle/ref_js/glsl2js.js + harness.js, run in the
build container by oracle/make_goldens.py) — see tests/test_oracle_golden.py and
tests/golden/.  The reference's repository holds no test or fixture for this path.

Array convention (same as the C-ABI): ``f[9][NY][NX]``, ``solid[NY][NX]`` uint8
(non-zero = solid), row 0 = bottom of the tunnel, x increasing downstream.
"""
from __future__ import annotations

import numpy as np

# html:238-248 (dir), html:249-253 (wt), html:254-264 (opp)
E = ((0, 0), (1, 0), (0, 1), (-1, 0), (0, -1), (1, 1), (-1, 1), (-1, -1), (1, -1))
OPP = (0, 3, 4, 1, 2, 7, 8, 5, 6)

# html:344 stability net
U_MAX, RHO_MIN, RHO_MAX = 0.35, 0.5, 2.0
# html:78, html:472, html:528
TAU_DEFAULT, U0_DEFAULT, VORT_SCALE = 0.58, 0.06, 0.06
# html:73
DX0, DX1 = -0.42, 1.42


def weights(T):
    """html:234-236: w0=4/9, ws=1/9, wd=1/36 evaluated in the working precision."""
    w0 = T(4.0) / T(9.0)
    ws = T(1.0) / T(9.0)
    wd = T(1.0) / T(36.0)
    return (w0, ws, ws, ws, ws, wd, wd, wd, wd)


def feq(i, rho, ux, uy, T):
    """html:276-281: wt(i)*rho*(1.0+3.0*eu+4.5*eu*eu-1.5*uu), literal order."""
    ex, ey = E[i]
    eu = T(ex) * ux + T(ey) * uy
    uu = ux * ux + uy * uy
    return (weights(T)[i] * rho) * (((T(1.0) + T(3.0) * eu) + (T(4.5) * eu) * eu) - T(1.5) * uu)


def equilibrium_init(nx, ny, u0, dtype):
    """html:474-490 equilibriumInitData: feq(rho=1,u=(u0,0)) evaluated in JS
    doubles, then stored (rounded) in the working precision; EVERY cell, solids
    included, gets it; macro = (1, u0, 0)."""
    T = np.dtype(dtype).type
    u0 = float(u0)
    w0, ws, wd = 4 / 9, 1 / 9, 1 / 36

    def feq64(ex, ey, w):
        eu = ex * u0
        uu = u0 * u0
        return w * (1 + 3 * eu + 4.5 * eu * eu - 1.5 * uu)

    vals = [feq64(0, 0, w0), feq64(1, 0, ws), feq64(0, 1, ws), feq64(-1, 0, ws), feq64(0, -1, ws),
            feq64(1, 1, wd), feq64(-1, 1, wd), feq64(-1, -1, wd), feq64(1, -1, wd)]
    f = np.empty((9, ny, nx), dtype=dtype)
    for i in range(9):
        f[i] = T(vals[i])
    rho = np.full((ny, nx), T(1.0), dtype=dtype)
    ux = np.full((ny, nx), T(u0), dtype=dtype)
    uy = np.zeros((ny, nx), dtype=dtype)
    return f, (rho, ux, uy)


def step(f, solid, tau, u0):
    """One pass of STEP_FS main() (html:283-360) over the whole lattice.

    Returns (f_out, (rho, ux, uy)).  Branch order — first match wins:
    solid (287-294) → outlet ix==NX-1 (301-312) → far field ix==0|iy==0|iy==NY-1
    (314-322) → interior pull-stream + half-way bounce-back + clamp + BGK (324-359).
    """
    T = f.dtype.type
    _, ny, nx = f.shape
    sol = solid != 0
    tau = T(tau)
    u0 = T(u0)
    fo = np.empty_like(f)
    rho_o = np.empty((ny, nx), dtype=f.dtype)
    ux_o = np.empty((ny, nx), dtype=f.dtype)
    uy_o = np.empty((ny, nx), dtype=f.dtype)

    # ---- interior fluid cells (html:324-359), on the [1:NY-1, 1:NX-1] view ----
    fin = []
    for i, (ex, ey) in enumerate(E):
        src = f[i, 1 - ey:ny - 1 - ey, 1 - ex:nx - 1 - ex]           # f_i(x - e_i)
        src_solid = sol[1 - ey:ny - 1 - ey, 1 - ex:nx - 1 - ex]       # mask(x - e_i)
        fin.append(np.where(src_solid, f[OPP[i], 1:ny - 1, 1:nx - 1], src))
    rho = fin[0]                                                       # 0.0 + fin[0]
    for i in range(1, 9):
        rho = rho + fin[i]
    with np.errstate(all="ignore"):
        ux = (fin[1] + fin[5] + fin[8] - fin[3] - fin[6] - fin[7]) / rho
        uy = (fin[2] + fin[5] + fin[6] - fin[4] - fin[7] - fin[8]) / rho
        rho = np.minimum(np.maximum(rho, T(RHO_MIN)), T(RHO_MAX))     # clamp (html:345)
        spd2 = ux * ux + uy * uy
        over = spd2 > T(U_MAX) * T(U_MAX)
        k = T(U_MAX) / np.sqrt(spd2)
        ux = np.where(over, ux * k, ux)
        uy = np.where(over, uy * k, uy)
        for i in range(9):
            eq = feq(i, rho, ux, uy, T)
            fo[i, 1:ny - 1, 1:nx - 1] = fin[i] - (fin[i] - eq) / tau
    rho_o[1:ny - 1, 1:nx - 1] = rho
    ux_o[1:ny - 1, 1:nx - 1] = ux
    uy_o[1:ny - 1, 1:nx - 1] = uy

    # ---- far field: inlet column, top and bottom rows (html:314-322) ----
    one = T(1.0)
    zero = T(0.0)
    for i in range(9):
        v = feq(i, one, u0, zero, T)
        fo[i, :, 0] = v
        fo[i, 0, :] = v
        fo[i, ny - 1, :] = v
    for arr, v in ((rho_o, one), (ux_o, u0), (uy_o, zero)):
        arr[:, 0] = v
        arr[0, :] = v
        arr[ny - 1, :] = v

    # ---- outlet column copies the un-streamed ix-1 populations (html:301-312);
    #      it wins over top/bottom at the two right-hand corners ----
    c = [f[i, :, nx - 2] for i in range(9)]
    r = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7] + c[8]
    with np.errstate(all="ignore"):
        oux = (c[1] + c[5] + c[8] - c[3] - c[6] - c[7]) / r
        ouy = (c[2] + c[5] + c[6] - c[4] - c[7] - c[8]) / r
    for i in range(9):
        fo[i, :, nx - 1] = c[i]
    rho_o[:, nx - 1] = r
    ux_o[:, nx - 1] = oux
    uy_o[:, nx - 1] = ouy

    # ---- solid cells: reversed populations, macro (1,0,0) (html:287-294) ----
    for i in range(9):
        fo[i][sol] = f[OPP[i]][sol]
    rho_o[sol] = one
    ux_o[sol] = zero
    uy_o[sol] = zero
    return fo, (rho_o, ux_o, uy_o)


def run(solid, steps, tau=TAU_DEFAULT, u0=U0_DEFAULT, dtype=np.float32, f=None):
    """initSim (html:492-500) followed by `steps` × simStep (html:510-525)."""
    ny, nx = solid.shape
    if f is None:
        f, macro = equilibrium_init(nx, ny, u0, dtype)
    else:
        macro = None
    for _ in range(steps):
        f, macro = step(f, solid, tau, u0)
    return f, macro


# --------------------------------------------------------------------------- #
# reductions (host-side JS in the reference: doubles on top of the fp32 macro)
# --------------------------------------------------------------------------- #
def ranges_from_macro(rho, ux, uy, solid, u0, prev=(0.6, -1.0, 1.0)):
    """html:596-614 updateFieldsFromMacro: returns (maxS, cpMin, cpMax).

    JS semantics: doubles; fluid cells only; maxS = max hypot(u,v) over values
    < 4 (strictly), cp range over -4 < cp < 1.2; a range keeps its previous
    value when no cell qualifies (html:611-613; initial values html:593)."""
    fluid = solid == 0
    u0 = float(u0)
    u = ux.astype(np.float64)[fluid] / u0
    v = uy.astype(np.float64)[fluid] / u0
    cp = (rho.astype(np.float64)[fluid] - 1) / (1.5 * u0 * u0)
    s = np.hypot(u, v)
    s = s[s < 4]
    mx = s.max() if s.size else 0.0
    q = cp[(cp > -4) & (cp < 1.2)]
    max_s = float(mx) if mx > 0 else prev[0]
    cp_min = float(q.min()) if q.size else prev[1]
    cp_max = float(q.max()) if q.size else prev[2]
    return max_s, cp_min, cp_max


def normalised_fields(rho, ux, uy, solid, u0):
    """html:600-606: Ufield=ux/U0, Vfield=uy/U0, CpField=(rho-1)/(1.5 U0^2),
    computed in doubles and stored as float32; NaN on solids."""
    u0 = float(u0)
    sol = solid != 0
    U = (ux.astype(np.float64) / u0).astype(np.float32)
    V = (uy.astype(np.float64) / u0).astype(np.float32)
    C = ((rho.astype(np.float64) - 1) / (1.5 * u0 * u0)).astype(np.float32)
    for a in (U, V, C):
        a[sol] = np.nan
    return U, V, C


def field_scalar(mode, rho, ux, uy, solid, u0, max_s, cp_min, cp_max, vort_scale=VORT_SCALE):
    """RENDER_FS main() field math (html:395-420), in the working precision:
    the scalar `t` handed to the colour map.  Solids → NaN (drawn flat, html:397).

    mode 0: t=(|u|/U0)/max(0.92*maxS,1e-6); mode 1: t=(cp-cpMin)/max(cpMax-cpMin,1e-6);
    mode 2: vort=(uy[x+1]-uy[x-1])*0.5-(ux[y+1]-ux[y-1])*0.5, t=vort/max(U0*vortScale,1e-6),
    CLAMP_TO_EDGE neighbours, solid neighbours contribute their stored (0,0)."""
    T = rho.dtype.type
    u0 = T(u0)
    if mode == 0:
        s = np.sqrt(ux * ux + uy * uy) / u0
        t = s / np.maximum(T(max_s) * T(0.92), T(1e-6))
    elif mode == 1:
        cp = (rho - T(1.0)) / (T(1.5) * u0 * u0)
        rng = np.maximum(T(cp_max) - T(cp_min), T(1e-6))
        t = (cp - T(cp_min)) / rng
    elif mode == 2:
        uyp = np.pad(uy, ((0, 0), (1, 1)), mode="edge")
        uxp = np.pad(ux, ((1, 1), (0, 0)), mode="edge")
        dvydx = (uyp[:, 2:] - uyp[:, :-2]) * T(0.5)
        duxdy = (uxp[2:, :] - uxp[:-2, :]) * T(0.5)
        vort = dvydx - duxdy
        t = vort / np.maximum(u0 * T(vort_scale), T(1e-6))
    else:
        raise ValueError("mode must be 0 (speed), 1 (cp) or 2 (vort)")
    t = t.astype(rho.dtype, copy=True)
    t[solid != 0] = np.nan
    return t


def compute_forces_raw(rho, ux, solid):
    """html:650-699 computeForces, raw sums: for every solid cell and each of its
    4 face neighbours that is inside the grid and fluid: p=rho_fluid/3 (double),
    fx += p*(-dx), fy += p*(-dy); surf counts such faces, rev those with ux<0.
    Returns (fx, fy, surf, rev).  Summation order differs from the JS scan
    (numerically irrelevant at double precision; tests use rtol 1e-12)."""
    sol = solid != 0
    ny, nx = sol.shape
    fx = 0.0
    fy = 0.0
    surf = 0
    rev = 0
    r64 = rho.astype(np.float64)
    for dx, dy in ((1, 0), (0, 1), (-1, 0), (0, -1)):
        # solid cell c=(y,x), neighbour (y+dy, x+dx) inside the grid and fluid
        ys = slice(max(0, -dy), ny - max(0, dy))
        xs = slice(max(0, -dx), nx - max(0, dx))
        yn = slice(max(0, -dy) + dy, ny - max(0, dy) + dy)
        xn = slice(max(0, -dx) + dx, nx - max(0, dx) + dx)
        face = sol[ys, xs] & ~sol[yn, xn]
        p = r64[yn, xn][face] / 3
        fx += float(np.sum(p * (-dx)))
        fy += float(np.sum(p * (-dy)))
        surf += int(face.sum())
        rev += int((ux[yn, xn][face] < 0).sum())
    return fx, fy, surf, rev


class ForceState:
    """html:641, 672-679, 699: CL/CD exponential smoothing (0.9/0.1, seeded with
    the first raw value) and separation fraction (0.85/0.15, seeded with 0)."""

    def __init__(self):
        self.cl = None
        self.cd = None
        self.sep = 0.0

    def update(self, fx, fy, surf, rev, u0, nx):
        if surf == 0:          # `if(!any) return;` html:672
            return
        chord_l = nx / (DX1 - DX0)                     # html:77
        q = 0.5 * u0 * u0 * chord_l
        cl_raw, cd_raw = fy / q, fx / q
        self.cl = cl_raw if self.cl is None else self.cl * 0.9 + cl_raw * 0.1
        self.cd = cd_raw if self.cd is None else self.cd * 0.9 + cd_raw * 0.1
        self.sep = self.sep * 0.85 + (rev / surf) * 0.15


def stall_label(sep_frac):
    """html:869-884."""
    pct = int(np.floor(sep_frac * 100 + 0.5))          # Math.round
    if pct < 5:
        return "Attached"
    if pct < 25:
        return f"{pct}% sep"
    return f"STALL ≈ {pct}% sep"


def lattice_reynolds(u0, nx, tau):
    """html:77-79, 865: Re = U0*CHORD_L/NU_L."""
    return u0 * (nx / (DX1 - DX0)) / ((tau - 0.5) / 3)
