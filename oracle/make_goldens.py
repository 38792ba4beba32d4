#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own code in this container.

TEST INFRASTRUCTURE.  Needs /root/reference and Node (present in the build
container only); the fixtures it writes are data and are committed, so neither
the tests nor the GPU box ever read /root/reference.

What is run (oracle/ref_js/harness.js):
  * the reference's pure-JS helpers, sliced out of
    pages/airfoil_flow_lbm_aerolab.html at run time with NX,NY overridden:
    naca4/clarkY/rotate/panelise/rasterMask/buildGeometry (html:99-182, 559-577),
    equilibriumInitData (html:474-490), updateFieldsFromMacro (html:596-614),
    computeForces (html:650-700), colour-map twins (html:704-719);
  * the reference's GLSL shaders STEP_FS (html:222-360) and RENDER_FS
    (html:362-422), transpiled from their text by oracle/ref_js/glsl2js.js and
    executed per lattice site in IEEE fp32 (or fp64) with one rounding per
    operation — i.e. the reference's step itself, headless.

usage: python oracle/make_goldens.py [--only NAME]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_HTML = "/root/reference/pages/airfoil_flow_lbm_aerolab.html"
HARNESS = os.path.join(HERE, "ref_js", "harness.js")
OUT = os.path.join(ROOT, "tests", "golden")


def run_harness(job: dict, tmp: str) -> dict:
    job = dict(job)
    job["outdir"] = tmp
    jp = os.path.join(tmp, "job.json")
    with open(jp, "w") as fh:
        json.dump(job, fh)
    res = subprocess.run(["node", "--max-old-space-size=6000", HARNESS, REF_HTML, jp], check=True, capture_output=True, text=True)
    return json.loads(res.stdout)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mask_spans(mask: np.ndarray) -> np.ndarray:
    """Run-length spans (iy, ix0, ix1) of the solid cells, row by row."""
    out = []
    for iy in range(mask.shape[0]):
        row = mask[iy] != 0
        if not row.any():
            continue
        d = np.diff(np.concatenate(([0], row.view(np.int8), [0])))
        for a, b in zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1)):
            out.append((iy, a, b - 1))
    return np.asarray(out, dtype=np.int32).reshape(-1, 3)


def open_te_user_coords(n=30, t=0.12):
    """A UIUC-style NACA 0012 with an OPEN trailing edge (-0.1015 coefficient), authored here,
    rounded to 6 dp as pages/Airfoil_Analysis.py:34-36 does before injection."""
    import math
    up, lo = [], []
    for i in range(n + 1):
        x = 0.5 * (1 - math.cos(math.pi * i / n))
        yt = 5 * t * (0.2969 * math.sqrt(x) - 0.126 * x - 0.3516 * x * x + 0.2843 * x ** 3 - 0.1015 * x ** 4)
        up.append([x, yt])
        lo.append([x, -yt])
    up.reverse()
    pts = up + lo[1:]
    return [[round(float(x), 6), round(float(y), 6)] for x, y in pts]


GEOMETRY_CASES = [
    # name, nx, ny, shape, aoa, dy_half (None = reference 0.46), user coords?
    ("geom_default_320x160_naca2412_a6", 320, 160, "naca2412", 6.0, None, False),
    ("geom_cfg1_256x128_naca0012_a0", 256, 128, "naca0012", 0.0, None, False),
    ("geom_cfg2_1024x512_naca2412_a5", 1024, 512, "naca2412", 5.0, None, False),
    ("geom_cfg5_4096x2048_naca4412_a12", 4096, 2048, "naca4412", 12.0, None, False),
    ("geom_refaxes_4096x4096_naca0012_a10", 4096, 4096, "naca0012", 10.0, None, False),
    ("geom_square_4096x4096_naca6409_a10", 4096, 4096, "naca6409", 10.0, 0.5 * (1.42 - -0.42) * 4096 / 4096, False),
    ("geom_cfg4_16384x4096_naca0012_a8", 16384, 4096, "naca0012", 8.0, 0.5 * (1.42 - -0.42) * 4096 / 16384, False),
    ("geom_clarky_200x131_am7p5", 200, 131, "clark_y", -7.5, 0.5 * (1.42 - -0.42) * 131 / 200, False),
    ("geom_naca6409_300x260_a10", 300, 260, "naca6409", 10.0, 0.5 * (1.42 - -0.42) * 260 / 300, False),
    ("geom_user_open_te_320x160_a4", 320, 160, "naca2412", 4.0, None, True),
    ("geom_neg_aoa_320x160_naca4412_am20", 320, 160, "naca4412", -20.0, None, False),
    ("geom_max_aoa_320x160_naca0012_a25", 320, 160, "naca0012", 25.0, None, False),
]

RUN_CASES = [
    # name, nx, ny, shape, aoa, mode, u0, tau, steps, extras
    dict(name="run_64x32_naca0012_a0_f32", nx=64, ny=32, shape="naca0012", aoa=0.0, mode="f32", u0=0.06, tau=None, steps=100, full=True, render=True, tracers=True),
    dict(name="run_64x32_naca0012_a0_f64", nx=64, ny=32, shape="naca0012", aoa=0.0, mode="f64", u0=0.06, tau=None, steps=100, full=True),
    dict(name="run_96x48_naca4412_a20_lowtau_f32", nx=96, ny=48, shape="naca4412", aoa=20.0, mode="f32", u0=0.10, tau=0.5004, steps=400, full=True),
    dict(name="run_96x48_sliders_f32", nx=96, ny=48, shape="naca2412", aoa=6.0, mode="f32", u0=0.06, tau=None, steps=120, full=True,
         aoa_schedule=[{"step": 40, "aoa": 14.5}, {"step": 80, "aoa": -3.0}], u0_schedule=[{"step": 60, "u0": 0.084}]),
    dict(name="run_default_320x160_naca2412_a6_f32", nx=320, ny=160, shape="naca2412", aoa=6.0, mode="f32", u0=0.06, tau=None, steps=200, full=False, tracers=True),
    dict(name="run_cfg1_256x128_naca0012_a0_f32", nx=256, ny=128, shape="naca0012", aoa=0.0, mode="f32", u0=0.06, tau=None, steps=500, full=False),
]


def tracer_points():
    """Probe positions for advect(): a lattice of points over (and a little beyond) the window,
    so that free-stream, near-body, in-body, edge-clamped and out-of-window cases all occur."""
    pts = []
    for a in range(-2, 40):
        for b in range(-2, 22):
            pts.append([-0.42 + 1.84 * (a + 0.37) / 37.0, -0.46 + 0.92 * (b + 0.41) / 19.0])
    pts += [[-0.42, 0.0], [1.42, 0.0], [0.3, -0.46], [0.3, 0.46], [-0.4199, 0.4599], [1.4199, -0.4599], [0.25, 0.0], [0.0, 0.0]]
    return pts


def gen_geometry(only=None):
    for name, nx, ny, shape, aoa, dy_half, user in GEOMETRY_CASES:
        if only and only != name:
            continue
        opts = {}
        if dy_half is not None:
            opts["dy_half"] = dy_half
        if user:
            opts["user_coords"] = open_te_user_coords()
        with tempfile.TemporaryDirectory() as tmp:
            res = run_harness({"nx": nx, "ny": ny, "opts": opts,
                               "geometry": {"shape": shape, "aoa": aoa, "mask_file": "mask.bin", "base_coords": True}}, tmp)
            mask = np.fromfile(os.path.join(tmp, "mask.bin"), dtype=np.uint8).reshape(ny, nx)
        g = res["geometry"]
        assert int((mask != 0).sum()) == g["solid_count"]
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            nx=nx, ny=ny, shape=shape, aoa=aoa, dy_half=-1.0 if dy_half is None else dy_half,
            user_coords=np.asarray(opts.get("user_coords", []), dtype=np.float64).reshape(-1, 2),
            base=np.asarray(g["base"], dtype=np.float64), xp=np.asarray(g["xp"]), yp=np.asarray(g["yp"]),
            solid_count=g["solid_count"], spans=mask_spans(mask), mask_sha256=sha(mask))
        print(f"{name}: solid={g['solid_count']}")


def gen_runs(only=None):
    for case in RUN_CASES:
        name = case["name"]
        if only and only != name:
            continue
        nx, ny, mode = case["nx"], case["ny"], case["mode"]
        dt = np.float32 if mode == "f32" else np.float64
        opts = {"init_f64": mode == "f64"}
        run = {"mode": mode, "u0": case["u0"], "steps": case["steps"], "state_file": "state.bin",
               "reduce": mode == "f32", "fields_file": "fields.bin"}
        if case.get("tau") is not None:
            run["tau"] = case["tau"]
        if case.get("render"):
            run["render_file"] = "render.bin"
        for k in ("aoa_schedule", "u0_schedule"):
            if case.get(k):
                run[k] = case[k]
        if case.get("tracers"):
            run["tracers"] = {"points": tracer_points(), "dt": 16.0}
        with tempfile.TemporaryDirectory() as tmp:
            res = run_harness({"nx": nx, "ny": ny, "opts": opts, "lattice": True, "init": {"u0": case["u0"]},
                               "geometry": {"shape": case["shape"], "aoa": case["aoa"], "mask_file": "mask.bin"},
                               "run": run}, tmp)
            n = nx * ny
            mask0 = np.fromfile(os.path.join(tmp, "mask.bin"), dtype=np.uint8).reshape(ny, nx)
            st = np.fromfile(os.path.join(tmp, "state.bin"), dtype=dt)
            f = st[:9 * n].reshape(9, ny, nx)
            rho, ux, uy = (st[(9 + k) * n:(10 + k) * n].reshape(ny, nx) for k in range(3))
            payload = dict(nx=nx, ny=ny, shape=case["shape"], aoa=case["aoa"], mode=mode, u0=case["u0"],
                           tau=res["consts"]["TAU"] if case.get("tau") is None else case["tau"], steps=case["steps"],
                           mask0_spans=mask_spans(mask0), rho=rho, ux=ux, uy=uy, f_sha256=sha(f),
                           lattice_e=np.asarray(res["lattice"]["e"], dtype=np.int32),
                           lattice_w=np.asarray(res["lattice"]["w"], dtype=np.float64),
                           lattice_opp=np.asarray(res["lattice"]["opp"], dtype=np.int32),
                           init_f=np.asarray(res["init"]["f"], dtype=np.float64),
                           schedules=json.dumps({k: case.get(k, []) for k in ("aoa_schedule", "u0_schedule")}))
            if case.get("full"):
                payload["f"] = f
            if mode == "f32":
                fl = np.fromfile(os.path.join(tmp, "fields.bin"), dtype=np.float32).reshape(3, ny, nx)
                payload.update(ranges=np.asarray([res["ranges"]["maxS"], res["ranges"]["cpMin"], res["ranges"]["cpMax"]]),
                               forces_first=np.asarray([res["forces"]["first"][k] for k in ("CLsmooth", "CDsmooth", "sepFrac")], dtype=np.float64),
                               forces_second=np.asarray([res["forces"]["second"][k] for k in ("CLsmooth", "CDsmooth", "sepFrac")], dtype=np.float64),
                               fields_sha256=sha(fl))
                if case.get("full"):
                    payload["fields"] = fl
            if case.get("tracers"):
                pts = np.asarray(tracer_points(), dtype=np.float64)
                adv = np.asarray([[np.nan] * 3 if a is None else a for a in res["tracers"]], dtype=np.float64)
                uv = np.asarray([[np.nan] * 2 if a is None else a for a in res["tracer_uv"]], dtype=np.float64)
                payload.update(tracer_points=pts, tracer_advect=adv, tracer_uv=uv, tracer_dt=16.0)
            if case.get("render"):
                payload["render_rgb"] = np.fromfile(os.path.join(tmp, "render.bin"), dtype=np.float32).reshape(3, ny, nx, 3)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
        print(f"{name}: {res['run']['seconds']:.1f}s in Node; rho in [{rho.min():.5f},{rho.max():.5f}]")


def gen_misc():
    with tempfile.TemporaryDirectory() as tmp:
        ts = [i / 40 for i in range(41)]
        res = run_harness({"nx": 320, "ny": 160, "cmaps": {"t": ts}, "init": {"u0": 0.06}, "lattice": True}, tmp)
        res2 = run_harness({"nx": 320, "ny": 160, "init": {"u0": 0.084}}, tmp)
    with open(os.path.join(OUT, "misc.json"), "w") as fh:
        json.dump({"consts_320x160": res["consts"], "init_u0_0.06": res["init"], "init_u0_0.084": res2["init"],
                   "lattice": res["lattice"], "cmap_t": ts, "cmaps": res["cmaps"]}, fh, indent=1)
    print("misc.json written")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-runs", action="store_true")
    args = ap.parse_args()
    if not os.path.exists(REF_HTML):
        sys.exit("reference not mounted: goldens can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    gen_geometry(args.only)
    if not args.skip_runs:
        gen_runs(args.only)
    if not args.only:
        gen_misc()


if __name__ == "__main__":
    main()
