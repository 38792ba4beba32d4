"""GPU: the two-steps-per-launch mode (csrc/step_march.hpp) must be bit-identical to the ordinary
single-step path and to the oracle: plain units (register-resident step 1 -> step 2), body units
(window-tile classes, bounce codes, inlet / outlet columns inside the march), odd/even step counts,
macro emission, mask changes, chunk sizes, the proved fast division by tau and its IEEE fallback; every
kernel form — fp32 (2 sites per lane, 128-row windows) and fp64 (1 site per lane, 64-row windows), each with two,
three and four steps per pass."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


# dtype, sites per lane (0: implied by the depth), steps per pass
COMBOS = (("float32", 2, 2), ("float32", 2, 3), ("float32", 2, 4), ("float64", 0, 2), ("float64", 1, 3), ("float64", 0, 4))


def _run(pkg, mask, chunks, tau, u0, fuse, chunk=None, sites=0, dtype="float32", depth=2):
    ny, nx = mask.shape
    with pkg.Engine(nx, ny, dtype=dtype) as e:
        e.set_option("fuse_steps", 0)
        if fuse:
            if chunk is not None:
                e.set_option("fuse_chunk", chunk)
            if sites:
                e.set_option("fuse_sites", sites)
            e.set_option("fuse_depth", depth)
            e.set_option("fuse_steps", 2)
        e.set_mask(mask)
        e.init_equilibrium(u0)
        if fuse:
            assert e.get_option("fuse_active") == 1.0 and e.get_option("fuse_units") > 0
            assert e.get_option("fuse_depth") == depth
            if sites:
                assert e.get_option("fuse_sites") == sites
        for n in chunks:
            e.step(n, tau, u0)
        return e.read_f(), e.read_macro(), e.info().steps_done


def _body(pkg, nx, ny, shape="naca2412", aoa=7.0):
    return pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask


@pytest.mark.parametrize("nx,ny,chunks,chunk", [
    (512, 256, [2], None),
    (512, 256, [1, 2, 3, 4, 5, 6, 17], None),
    (512, 256, [40], 1),
    (512, 256, [40], 7),
    (512, 256, [41], 100),
    (1024, 512, [64, 3], None),
    (384, 768, [30], 16),            # three windows per column
    (200, 1024, [21], 5),            # narrow, tall: five windows, body crosses window seams
    (2048, 1024, [12], None),
])
def test_fused_equals_single_step(pkg, nx, ny, chunks, chunk):
    mask = _body(pkg, nx, ny)
    ref = {}
    for dtype, sites, depth in COMBOS:
        if dtype not in ref:
            ref[dtype] = _run(pkg, mask, chunks, 0.58, 0.06, False, dtype=dtype)
        f0, m0, n0 = ref[dtype]
        f1, m1, n1 = _run(pkg, mask, chunks, 0.58, 0.06, True, chunk, sites, dtype, depth)
        assert n0 == n1 == sum(chunks)
        assert bits_equal(f0, f1), (dtype, sites, depth)
        assert all(bits_equal(a, b) for a, b in zip(m0, m1)), (dtype, sites, depth)


def test_fused_vs_oracle_and_edge_masks(pkg, oracle_c):
    nx, ny = 512, 512
    empty = np.zeros((ny, nx), np.uint8)
    wall = np.zeros((ny, nx), np.uint8); wall[:, 200:204] = 1
    specks = np.zeros((ny, nx), np.uint8); specks[::37, ::53] = 1; specks[255:257, 100:110] = 1; specks[251:254, 300] = 1
    edges = np.zeros((ny, nx), np.uint8); edges[0, 10:20] = 1; edges[ny - 1, 30:40] = 1; edges[50:60, 0] = 1; edges[70:90, nx - 1] = 1; edges[100:140, 2] = 1
    specks[127:129, 400:410] = 1; specks[383:386, 450] = 1          # across the seams of 128-row windows too
    for mask in (empty, wall, specks, edges, _body(pkg, nx, ny, "naca4412", 15.0)):
        ref = {}
        for dtype, sites, depth in COMBOS:
            if dtype not in ref:
                ref[dtype] = oracle_c.run(mask, 17, 0.58, 0.06, np.dtype(dtype))
            fr, mr = ref[dtype]
            f, m, _ = _run(pkg, mask, [8, 9], 0.58, 0.06, True, 12, sites, dtype, depth)
            assert bits_equal(f, fr) and all(bits_equal(a, b) for a, b in zip(m, mr)), (dtype, sites, depth)


def test_fused_low_tau_clamp_and_mask_change(pkg, oracle_c):
    nx, ny = 512, 256
    m1, m2 = _body(pkg, nx, ny, "naca4412", 20.0), _body(pkg, nx, ny, "naca4412", 5.0)
    fr, _ = oracle_c.run(m1, 300, 0.5004, 0.10, np.float32)
    fr, mr = oracle_c.run(m2, 100, 0.5004, 0.09, np.float32, f=fr)
    for sites, depth in ((2, 2), (0, 3), (2, 4)):
        with pkg.Engine(nx, ny) as e:
            e.set_option("fuse_sites", sites)
            e.set_option("fuse_depth", depth)
            e.set_option("fuse_steps", 2)
            e.set_mask(m1); e.init_equilibrium(0.10); e.step(300, 0.5004, 0.10)
            e.set_mask(m2); e.step(100, 0.5004, 0.09)
            f, m = e.read_f(), e.read_macro()
            events = e.clamp_events()
        assert bits_equal(f, fr) and all(bits_equal(a, b) for a, b in zip(m, mr)), (sites, depth)
    fr64, _ = oracle_c.run(m1, 300, 0.5004, 0.10, np.float64)
    fr64, mr64 = oracle_c.run(m2, 100, 0.5004, 0.09, np.float64, f=fr64)
    with pkg.Engine(nx, ny, dtype="float64") as e:
        e.set_option("fuse_steps", 2)
        e.set_mask(m1); e.init_equilibrium(0.10); e.step(300, 0.5004, 0.10)
        e.set_mask(m2); e.step(100, 0.5004, 0.09)
        assert e.get_option("fuse_active") == 1.0
        assert bits_equal(e.read_f(), fr64) and all(bits_equal(a, b) for a, b in zip(e.read_macro(), mr64))
    # the clamp-event diagnostic (html:344-350) counted on the oracle's macro state
    fluid = m2 == 0
    rho, ux, uy = (a.astype(np.float64) for a in mr)
    want = (int(((mr[0] == np.float32(0.5)) | (mr[0] == np.float32(2.0)))[fluid].sum()),
            int(((ux * ux + uy * uy) >= 0.35 * 0.35 * (1 - 1e-6))[fluid].sum()))
    assert events == want and want[1] > 0                      # this run does hit the speed clamp
    with pkg.Engine(nx, ny) as e:
        e.set_mask(m2); e.init_equilibrium(0.06); e.step(50, 0.58, 0.06)
        assert e.clamp_events() == (0, 0)                       # a healthy run never touches the net


def test_fused_toggle_midrun_and_4096(pkg):
    nx = ny = 4096
    mask = _body(pkg, nx, ny, "naca6409", 10.0)
    with pkg.Engine(nx, ny) as a, pkg.Engine(nx, ny) as b:
        for e in (a, b):
            e.set_mask(mask); e.init_equilibrium(0.06)
        a.step(9, 0.58, 0.06)
        b.step(3, 0.58, 0.06)
        b.set_option("fuse_steps", 2)            # option set after the mask: the plan is rebuilt from the kept copy
        assert b.get_option("fuse_active") == 1.0
        b.step(4, 0.58, 0.06)
        assert b.get_option("fast_div_active") == 1.0
        assert 0 < b.get_option("fuse_tiles_general") < 0.15 * nx * (ny // 256)      # window-tiles that take the body paths
        assert b.get_option("fuse_units") <= 2 * 256 * 8                             # whole resident rounds of units
        b.set_option("fuse_steps", 0)
        b.step(2, 0.58, 0.06)
        assert bits_equal(a.read_f(), b.read_f())
        assert all(bits_equal(x, y) for x, y in zip(a.read_macro(), b.read_macro()))


def test_fused_not_available(pkg):
    with pkg.Engine(256, 128, dtype="float64") as e:
        with pytest.raises(pkg.WTError):
            e.set_option("fuse_sites", 2)           # fp64: one site per lane
        with pytest.raises(pkg.WTError):
            e.set_option("fuse_depth", 5)
        e.set_option("fuse_steps", 2)
        e.set_mask(np.zeros((128, 256), np.uint8)); e.init_equilibrium(0.06); e.step(7, 0.58, 0.06)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("fuse_depth") == 3 and e.get_option("fuse_sites") == 1    # fp64: one site per lane
        e.set_option("fuse_depth", 2)
        e.step(5, 0.58, 0.06)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("fuse_depth") == 2 and e.get_option("fuse_sites") == 1
        assert e.get_option("single_steps") == 1.0      # 7 = 3 + 2 + 2, then 5 = 2 + 2 + 1
    with pkg.Engine(12, 64) as e:                   # fewer than 16 columns: two steps per pass at most
        with pytest.raises(pkg.WTError):
            e.set_option("fuse_depth", 3)
        e.set_option("fuse_steps", 2)
        e.set_mask(np.zeros((64, 12), np.uint8)); e.init_equilibrium(0.06); e.step(5, 0.58, 0.06)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("fuse_depth") == 2
    with pkg.Engine(256, 129) as e:                 # odd NY: no vector width divides it
        with pytest.raises(pkg.WTError):
            e.set_option("fuse_steps", 2)
        with pytest.raises(pkg.WTError):
            e.set_option("no_such_option", 1)
    with pkg.Engine(256, 130) as e:                 # fp32: two sites per lane, nothing else
        with pytest.raises(pkg.WTError):
            e.set_option("fuse_sites", 4)           # the 16-byte-vector kernels of round 2 are gone
        with pytest.raises(pkg.WTError):
            e.set_option("fuse_sites", 3)
        e.set_option("fuse_steps", 2)
        e.set_mask(np.zeros((130, 256), np.uint8)); e.init_equilibrium(0.06); e.step(4, 0.58, 0.06)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("fuse_sites") == 2
    with pkg.Engine(64, 64) as e:               # eligible but tiny: a plan with few units still works
        e.set_option("fuse_steps", 2)
        e.set_mask(np.zeros((64, 64), np.uint8)); e.init_equilibrium(0.06); e.step(6, 0.58, 0.06)
        assert e.info().steps_done == 6


def test_long_run_fused_equals_single_step_and_stays_finite(pkg):
    """5000 steps on the bench lattice (4096^2, NACA 6409, 10 deg): both modes end in the same bits;
    and the near-stall fp32 case at tau ~ 0.5004 (clamp active) stays finite over 3000 steps."""
    import hashlib
    nx = ny = 4096
    mask = _body(pkg, nx, ny, "naca6409", 10.0)
    digests = []
    for fuse in (0, 1):
        with pkg.Engine(nx, ny) as e:
            e.set_option("fuse_steps", 2 * fuse)
            e.set_mask(mask); e.init_equilibrium(0.06)
            for _ in range(5):
                e.step(1000, 0.58, 0.06)
            rho, ux, uy = e.read_macro()
            assert np.isfinite(rho).all() and np.isfinite(ux).all() and 0.9 < float(rho.mean()) < 1.1
            digests.append(hashlib.sha256(rho.tobytes() + ux.tobytes() + uy.tobytes()).hexdigest())
    assert digests[0] == digests[1]
    with pkg.WindTunnel(shape="naca4412", nx=2048, ny=1024, aoa_deg=12.0, re=1e6 * 2048 / 4096) as wt:
        wt.engine.set_option("fuse_steps", 2)
        wt.sim_step(3000)
        rho, ux, uy = wt.read_macro()
        assert np.isfinite(rho).all() and np.isfinite(ux).all() and np.isfinite(uy).all()
        assert float(rho.min()) >= 0.5 and float(rho.max()) <= 2.0
        assert float(np.hypot(ux, uy).max()) <= 0.35 * (1 + 1e-6)          # html:344-350 clamp bounds


@pytest.mark.parametrize("nranks,halo,nx,ny,chunks,dtype,sites,depth", [
    (2, 4, 512, 256, [1, 2, 3, 8, 21], "float32", 2, 2),
    (3, 7, 768, 512, [40], "float32", 0, 2),
    (2, 1, 512, 256, [9], "float32", 2, 2),              # halo 1: never two exact ghost columns -> single steps only
    (8, 16, 4096, 256, [50], "float32", 2, 2),
    (4, 16, 2048, 512, [33, 18], "float32", 2, 2),
    (3, 7, 768, 512, [40], "float64", 0, 2),
    (2, 4, 512, 256, [1, 2, 3, 8, 21], "float32", 2, 3),
    (3, 7, 768, 512, [40], "float32", 2, 3),
    (3, 7, 768, 512, [40], "float64", 0, 3),
    (4, 17, 2048, 512, [33, 18], "float32", 2, 4),
    (2, 4, 512, 256, [1, 2, 3, 8, 21], "float32", 2, 4),
    (3, 9, 768, 512, [40], "float64", 0, 4),
    (2, 3, 512, 256, [9], "float32", 2, 4),              # halo 3: never four exact ghost columns -> shorter passes and single steps
    (4, 17, 2048, 512, [33, 18], "float32", 2, 3),
    (2, 2, 512, 256, [9], "float32", 2, 3),              # halo 2: never three exact ghost columns -> single steps only
    (8, 16, 4096, 256, [50], "float32", 0, 0),           # automatic choice
    (2, 2, 2304, 4096, [9, 6], "float32", 0, 0),         # wide slabs would plan four steps per pass, but two ghost columns allow three at most
    (2, 1, 2304, 4096, [5], "float32", 0, 0),            # one ghost column: the two-step kernel's tables (its units leave one column), no fused pass fits
])
def test_fused_slabs_equal_single_lattice(pkg, nranks, halo, nx, ny, chunks, dtype, sites, depth):
    """Two-steps-per-launch on column slabs (in-process transport): a pair needs two exact ghost
    columns, refresh steps stay single; results equal the plain single lattice bit for bit."""
    mask = _body(pkg, nx, ny, "naca2412", 7.0)
    f0, m0, _ = _run(pkg, mask, chunks, 0.58, 0.06, False, dtype=dtype)
    es = [pkg.Engine(nx, ny, rank=r, nranks=nranks, halo=halo, dtype=dtype) for r in range(nranks)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            if sites:
                e.set_option("fuse_sites", sites)
            if depth:
                e.set_option("fuse_depth", depth)
            e.set_option("fuse_steps", 2)
            e.set_mask(mask); e.init_equilibrium(0.06)
        assert any(e.get_option("fuse_active") == 1.0 for e in es)
        assert not depth or all(e.get_option("fuse_depth") == depth for e in es)
        for n in chunks:
            pkg.Engine.step_group(es, n, 0.58, 0.06)
        f1 = np.concatenate([e.read_f() for e in es], axis=2)
        m1 = [np.concatenate(p, axis=1) for p in zip(*[e.read_macro() for e in es])]
        assert all(e.info().steps_done == sum(chunks) for e in es)
    finally:
        for e in es:
            e.close()
    assert bits_equal(f0, f1)
    assert all(bits_equal(a, b) for a, b in zip(m0, m1))


def test_fuse_auto_only_where_it_pays(pkg):
    """fuse_steps = 1 engages only when the plan has at least as many units as resident wave slots."""
    with pkg.Engine(512, 256) as small, pkg.Engine(4096, 1024) as big:
        for e in (small, big):
            e.set_option("fuse_steps", 1)
            e.set_mask(np.zeros((e.ny, e.nx_global), np.uint8)); e.init_equilibrium(0.06)
        assert small.get_option("fuse_active") == 0.0 and big.get_option("fuse_active") == 1.0
        assert big.get_option("fuse_sites") == 2
        small.set_option("fuse_steps", 2)
        assert small.get_option("fuse_active") == 1.0
        small.step(6, 0.58, 0.06); big.step(6, 0.58, 0.06)
    with pkg.Engine(4096, 4096) as wide, pkg.Engine(544, 4096) as slab, pkg.Engine(4096, 2048, dtype="float64") as f64:
        for e in (wide, slab, f64):
            e.set_mask(np.zeros((e.ny, e.nx_global), np.uint8)); e.init_equilibrium(0.06)
            assert e.get_option("fuse_active") == 1.0     # the default
        assert [(e.get_option("fuse_depth"), e.get_option("fuse_sites")) for e in (wide, slab, f64)] == [(4, 2), (4, 2), (4, 1)]
        for e in (wide, slab, f64):
            e.set_option("fuse_depth", 2)
        assert [(e.get_option("fuse_depth"), e.get_option("fuse_sites")) for e in (wide, slab, f64)] == [(2, 2), (2, 2), (2, 1)]


def test_set_mask_stays_interactive(pkg):
    """The AoA slider (html:943-947, 35 ms debounce) re-uploads the mask: upload, tile classes, bounce codes and the
    marching plan together must stay far below a frame (<= 5 ms at 1024x512)."""
    import time
    for (nx, ny), limit_ms in (((1024, 512), 5.0), ((4096, 4096), 25.0)):
        masks = [_body(pkg, nx, ny, "naca2412", a) for a in (4.0, 4.5, 5.0, 5.5)]
        with pkg.Engine(nx, ny) as e:
            e.set_option("fuse_steps", 2)
            e.set_mask(masks[0]); e.init_equilibrium(0.06); e.step(4, 0.58, 0.06); e.sync()
            ts = []
            for k in range(12):
                t0 = time.perf_counter(); e.set_mask(masks[k % 4]); ts.append((time.perf_counter() - t0) * 1e3)
            assert e.get_option("fuse_active") == 1.0
            assert sorted(ts)[len(ts) // 2] <= limit_ms, ts
            e.step(6, 0.58, 0.06)


def test_remainder_of_one_never_falls_back_to_single_steps(pkg, oracle_c):
    """fuse_stride: step(5), step(9), step(13) on a depth-4 plan split as 3 + 2, 4 + 3 + 2, 4 + 4 + 3 + 2 — never a pass plus a single
    k_step, which would clear the seam buffer and send the next pass through the halo kernels' gather path.  Same for 4 = 2 + 2 and
    7 = 3 + 2 + 2 on a depth-3 plan."""
    nx, ny = 4096, 1024
    mask = _body(pkg, nx, ny, "naca2412", 6.0)
    fr, mr = oracle_c.run(mask, 27, 0.58, 0.06, np.float32)
    with pkg.Engine(nx, ny) as e:
        e.set_option("fuse_depth", 4)
        e.set_mask(mask); e.init_equilibrium(0.06)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("fuse_depth") == 4
        for n, passes in ((5, 2), (9, 3), (13, 4)):
            p0 = e.get_option("passes")
            e.step(n, 0.58, 0.06)
            assert e.get_option("passes") - p0 == passes
        assert e.get_option("single_steps") == 0
        assert bits_equal(e.read_f(), fr) and all(bits_equal(a, b) for a, b in zip(e.read_macro(), mr))
    with pkg.Engine(nx, ny) as e:
        e.set_option("fuse_depth", 3)
        e.set_mask(mask); e.init_equilibrium(0.06)
        for n in (4, 7, 4, 5, 7):
            e.step(n, 0.58, 0.06)
        assert e.get_option("single_steps") == 0 and e.info().steps_done == 27
        assert bits_equal(e.read_f(), fr)


def test_slabs_choose_one_plan_depth_and_mixed_groups_still_take_fused_passes(pkg):
    """Every slab of a tunnel must take the same sequence of passes and refresh steps (over RCCL each rank decides alone, and the exchange is
    collective), so the automatic steps per pass come from the NARROWEST slab of the split: here an edge slab (W + halo = 316 columns, 4.9 per
    resident unit -> three steps per pass) decides for the interior slabs too (332 columns, 5.1 per unit: alone they would plan four).  And when the depths DO
    differ (forced per handle — ADVICE r2), wt_step_group runs passes of the length every slab can take and refreshes as soon as any slab has
    run out of exact ghost columns: a three-step pass on four-step tables still costs the three unwritten columns next to a local edge."""
    nranks, halo, nx, ny = 4, 16, 4 * 300, 4096
    mask = _body(pkg, nx, ny, "naca2412", 7.0)
    steps = [17, 16]
    with pkg.Engine(nx, ny) as ref:
        ref.set_option("fuse_steps", 0)
        ref.set_mask(mask); ref.init_equilibrium(0.06)
        for n in steps:
            ref.step(n, 0.58, 0.06)
        f0 = ref.read_f()
    for forced in (False, True):
        es = [pkg.Engine(nx, ny, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
        try:
            pkg.Engine.link_local(es)
            for r, e in enumerate(es):
                if forced:
                    e.set_option("agree_check", 0)       # mixed on purpose: the library would otherwise refuse the group (test_gpu_slabs.py)
                if forced and 0 < r < nranks - 1:
                    e.set_option("fuse_depth", 4)
                e.set_mask(mask); e.init_equilibrium(0.06)
            depths = [int(e.get_option("fuse_depth")) for e in es]
            assert all(e.get_option("fuse_active") == 1.0 for e in es)
            assert depths == ([3, 4, 4, 3] if forced else [3, 3, 3, 3]), depths
            for n in steps:
                pkg.Engine.step_group(es, n, 0.58, 0.06)
            # 33 steps with a few ghost refreshes: everything else went through fused passes on every slab
            assert all(e.get_option("passes") >= 8 and e.get_option("single_steps") <= 8 for e in es), [(e.get_option("passes"), e.get_option("single_steps")) for e in es]
            f1 = np.concatenate([e.read_f() for e in es], axis=2)
        finally:
            for e in es:
                e.close()
        assert bits_equal(f0, f1), forced


@pytest.mark.parametrize("dtype,depth", [("float32", 4), ("float32", 3), ("float64", 4)])
def test_measured_refinement_of_the_units_changes_no_bit(pkg, oracle_c, dtype, depth):
    """The library times the units of a new plan and cuts the columns again (tune_fuse_plan) before its first pass: the trial passes
    must leave the populations, the step count and (rho,ux,uy) alone, and a refined plan computes the bits of the modelled one and
    of the oracle."""
    nx, ny, tau, u0 = 1536, 1024, 0.56, 0.07
    mask = _body(pkg, nx, ny)
    out = {}
    for tune in (1, 0):
        with pkg.Engine(nx, ny, dtype=dtype) as e:
            e.set_option("tune", tune)
            e.set_option("fuse_depth", depth)
            e.set_option("fuse_steps", 2)
            e.set_mask(mask)
            e.init_equilibrium(u0)
            e.step(1, tau, u0)                      # a single step: no plan is timed for it
            assert e.get_option("tune_rounds") == 0
            f1 = e.read_f()
            e.step(depth, tau, u0)                  # the first pass: the plan is timed before it
            assert (e.get_option("tune_rounds") > 0) == bool(tune)
            assert e.get_option("tune_gain") >= 1.0 or not tune
            assert e.info().steps_done == 1 + depth
            e.step(2 * depth + 1, tau, u0)
            out[tune] = (f1, e.read_f(), e.read_macro(), e.get_option("fuse_units"))
            # a mask that follows a short-lived one (a slider being dragged: fewer than 16 passes) is timed only once it has lived that long
            # itself; a mask that follows a long-lived one is timed at once; turning the option off brings the modelled cut back
            e.set_mask(_body(pkg, nx, ny, aoa=3.0))
            e.step(depth, tau, u0)
            assert e.get_option("tune_rounds") == 0
            e.step(16 * depth, tau, u0)
            assert e.get_option("tune_rounds") == 0              # (looked at when a stepping call begins)
            e.step(depth, tau, u0)
            assert (e.get_option("tune_rounds") > 0) == bool(tune)
            e.set_mask(_body(pkg, nx, ny, aoa=4.0))
            e.step(depth, tau, u0)
            assert (e.get_option("tune_rounds") > 0) == bool(tune)
            e.set_option("tune", 0)
            e.step(depth, tau, u0)
            assert e.get_option("tune_rounds") == 0
    assert bits_equal(out[1][0], out[0][0]) and bits_equal(out[1][1], out[0][1])
    for a, b in zip(out[1][2], out[0][2]):
        assert bits_equal(a, b)
    fr, mr = oracle_c.run(mask, 1 + depth + 2 * depth + 1, tau, u0, np.dtype(dtype))
    assert bits_equal(out[1][1], fr)
    for a, b in zip(out[1][2], mr):
        assert bits_equal(a, b)
