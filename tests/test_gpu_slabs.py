"""GPU: column-slab decomposition inside libwindtunnel (ghost columns, deep halos, overlap of the
halo refresh with the interior columns) against the single-lattice run.  All slabs live on the one
GPU of the test box and move ghost columns with peer copies (wt_link_local / wt_step_group); the
RCCL transport shares every line of this logic except the send/recv calls themselves."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def _single(pkg, mask, chunks, tau, u0, dtype):
    ny, nx = mask.shape
    with pkg.Engine(nx, ny, dtype=dtype) as e:
        e.set_mask(mask); e.init_equilibrium(u0)
        for n in chunks:
            e.step(n, tau, u0)
        return e.read_f(), e.read_macro(), e.reduce_ranges(u0), e.forces()


def _slabs(pkg, mask, nranks, halo, chunks, tau, u0, dtype, edges=None):
    ny, nx = mask.shape
    es = [pkg.Engine(nx, ny, dtype=dtype, rank=r, nranks=nranks, halo=halo, edges=edges) for r in range(nranks)]
    try:
        if edges is not None:
            assert [e.x0 for e in es] == list(edges[:-1]) and [e.width for e in es] == [b - a for a, b in zip(edges[:-1], edges[1:])]
        assert sum(e.width for e in es) == nx and es[0].x0 == 0
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.init_equilibrium(u0)
        for n in chunks:
            pkg.Engine.step_group(es, n, tau, u0)
        f = np.concatenate([e.read_f() for e in es], axis=2)
        macro = [np.concatenate(parts, axis=1) for parts in zip(*[e.read_macro() for e in es])]
        rr = [e.reduce_ranges(u0) for e in es]
        ranges = (max(r[0] for r in rr), min(r[1] for r in rr), max(r[2] for r in rr))
        ff = [e.forces() for e in es]
        forces = (sum(x[0] for x in ff), sum(x[1] for x in ff), sum(x[2] for x in ff), sum(x[3] for x in ff))
        return f, macro, ranges, forces
    finally:
        for e in es:
            e.close()


@pytest.mark.parametrize("nranks,halo,nx,ny,dtype", [
    (2, 1, 512, 256, "float32"),
    (2, 4, 512, 256, "float32"),
    (4, 3, 512, 256, "float32"),
    (3, 2, 333, 200, "float32"),      # uneven slabs, ragged tiles
    (8, 16, 1024, 256, "float32"),    # the bench configuration's shape in miniature
    (4, 5, 512, 128, "float64"),
])
def test_slabs_equal_single_lattice(pkg, nranks, halo, nx, ny, dtype):
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask     # body spans several slabs
    chunks = [1, 2, halo, 2 * halo + 1, 23]
    f0, m0, r0, F0 = _single(pkg, mask, chunks, 0.58, 0.06, dtype)
    f1, m1, r1, F1 = _slabs(pkg, mask, nranks, halo, chunks, 0.58, 0.06, dtype)
    assert bits_equal(f0, f1)
    assert all(bits_equal(a, b) for a, b in zip(m0, m1))
    assert r0 == r1
    assert F0[2:] == F1[2:]
    np.testing.assert_allclose(F0[:2], F1[:2], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("edges,halo,nx,ny,dtype,fuse", [
    ([0, 100, 130, 400, 512], 3, 512, 256, "float32", 0),
    ([0, 700, 1100, 1500, 2048], 16, 2048, 1024, "float32", 2),       # marching passes; the widest slab alone would plan deeper than the group does
    ([0, 300, 330, 1024], 8, 1024, 512, "float64", 2),
    ([0, 40, 2048], 16, 2048, 1024, "float32", 2),
])
def test_slabs_cut_by_the_caller_equal_single_lattice(pkg, edges, halo, nx, ny, dtype, fuse):
    """wt_create_slab_at: slabs of unequal widths (the host cuts them by measured cost, distributed.balance_split).  Same bits as the
    whole lattice; every slab takes the schedule the narrowest one allows."""
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask
    chunks = [1, 2, halo, 2 * halo + 1, 23]
    f0, m0, r0, F0 = _single(pkg, mask, chunks, 0.58, 0.06, dtype)
    f1, m1, r1, F1 = _slabs(pkg, mask, len(edges) - 1, halo, chunks, 0.58, 0.06, dtype, edges=edges)
    assert bits_equal(f0, f1)
    assert all(bits_equal(a, b) for a, b in zip(m0, m1))
    assert r0 == r1 and F0[2:] == F1[2:]
    if fuse:
        es = [pkg.Engine(nx, ny, dtype=dtype, rank=r, nranks=len(edges) - 1, halo=halo, edges=edges) for r in range(len(edges) - 1)]
        try:
            for e in es:
                e.set_mask(mask); e.init_equilibrium(0.06)
            plans = [tuple(e.plan_steps(41, 0.58)) for e in es]
            assert len(set(plans)) == 1, plans
            assert len({(e.get_option("fuse_active"), e.get_option("fuse_depth")) for e in es}) == 1
        finally:
            for e in es:
                e.close()


def test_slab_split_errors(pkg):
    for bad in ([0, 10, 10, 64], [1, 32, 64], [0, 32, 60], [0, 40, 30, 64]):
        with pytest.raises(pkg.WTError):
            pkg.Engine(64, 64, rank=0, nranks=len(bad) - 1, halo=1, edges=bad)
    with pytest.raises(pkg.WTError):
        pkg.Engine(64, 64, rank=0, nranks=2, halo=8, edges=[0, 60, 64])        # halo wider than the narrowest slab
    with pytest.raises(ValueError):
        pkg.Engine(64, 64, rank=0, nranks=2, halo=1, edges=[0, 64])


def test_slab_restart_from_written_state(pkg):
    """wt_write_f invalidates the ghosts: the next step must refresh them first."""
    rng = np.random.default_rng(7)
    nx, ny = 512, 256
    mask = pkg.geometry.build_geometry(nx, ny, 3.0, None, "naca0012").mask
    f_init = (0.1 + 0.01 * rng.random((9, ny, nx))).astype(np.float32)
    with pkg.Engine(nx, ny) as e:
        e.set_mask(mask); e.write_f(f_init); e.step(9, 0.58, 0.06)
        want = e.read_f()
    es = [pkg.Engine(nx, ny, rank=r, nranks=2, halo=2) for r in range(2)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.write_f(np.ascontiguousarray(f_init[:, :, e.x0:e.x0 + e.width]))
        pkg.Engine.step_group(es, 9, 0.58, 0.06)
        got = np.concatenate([e.read_f() for e in es], axis=2)
    finally:
        for e in es:
            e.close()
    assert bits_equal(want, got)


def test_slab_argument_errors(pkg):
    with pytest.raises(pkg.WTError):
        pkg.Engine(64, 64, rank=0, nranks=2, halo=0)
    with pytest.raises(pkg.WTError):
        pkg.Engine(64, 64, rank=0, nranks=2, halo=40)         # wider than the slab
    e = pkg.Engine(64, 64, rank=0, nranks=2, halo=2)
    try:
        e.set_mask(np.zeros((64, 64), np.uint8)); e.init_equilibrium(0.06)
        with pytest.raises(pkg.WTError):
            e.step(1, 0.58, 0.06)                              # no transport yet
    finally:
        e.close()


def test_slab_fields_equal_single_lattice(pkg):
    """wt_field on slabs (vorticity pulls the neighbours' uy edge columns) vs the single lattice."""
    nx, ny = 512, 256
    mask = pkg.geometry.build_geometry(nx, ny, 12.0, None, "naca4412").mask
    with pkg.Engine(nx, ny) as e:
        e.set_mask(mask); e.init_equilibrium(0.06); e.step(90, 0.58, 0.06)
        rng = e.reduce_ranges(0.06)
        want = [e.field(m, 0.06, rng[0], rng[1], rng[2], 0.06) for m in range(3)]
    es = [pkg.Engine(nx, ny, rank=r, nranks=4, halo=3) for r in range(4)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.init_equilibrium(0.06)
        pkg.Engine.step_group(es, 90, 0.58, 0.06)
        for m in range(3):
            got = np.concatenate([e.field(m, 0.06, rng[0], rng[1], rng[2], 0.06) for e in es], axis=1)
            assert bits_equal(np.nan_to_num(got), np.nan_to_num(want[m])) and np.array_equal(np.isnan(got), np.isnan(want[m]))
    finally:
        for e in es:
            e.close()


def test_slab_wind_tunnel_with_real_engine_world1(pkg, oracle_c):
    """The torch.distributed host class (SlabWindTunnel) over the real engine, world_size 1 (gloo
    group created in-process): same results as WindTunnel / the oracle."""
    import os
    import socket
    import torch.distributed as dist
    from airfoil_cfd_tool_amd.distributed import SlabWindTunnel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with SlabWindTunnel(shape="naca2412", nx=512, ny=256, aoa_deg=6.0, device=0) as wt:
            for _ in range(6):
                wt.frame()
            rho, ux, uy = wt.read_macro()
            f = wt.read_f()
            t = wt.render_field(field="vort")
            assert wt.render_rgba().shape == (256, 512, 4)
            fr, mr = oracle_c.run(wt.geometry.mask, 24, 0.58, 0.06, np.float32)
            assert bits_equal(f, fr) and bits_equal(rho, mr[0]) and bits_equal(ux, mr[1]) and bits_equal(uy, mr[2])
            assert t.shape == (256, 512) and wt.stats().cl is not None
    finally:
        dist.destroy_process_group()


def test_local_slab_wind_tunnel_one_process(pkg, oracle_c):
    """One process driving several slab handles (here all on GPU 0): the WindTunnel surface, unchanged."""
    from airfoil_cfd_tool_amd.distributed import LocalSlabWindTunnel
    with LocalSlabWindTunnel(shape="naca4412", nx=768, ny=256, aoa_deg=9.0, devices=[0, 0, 0], halo=5) as wt, \
            pkg.WindTunnel(shape="naca4412", nx=768, ny=256, aoa_deg=9.0) as ref:
        for _ in range(7):
            wt.frame(); ref.frame()
        wt.aoa_deg = 13.0; ref.aoa_deg = 13.0
        wt.sim_step(9); ref.sim_step(9)
        assert bits_equal(wt.read_f(), ref.read_f())
        assert all(bits_equal(a, b) for a, b in zip(wt.read_macro(), ref.read_macro()))
        assert wt.update_fields_from_macro() == ref.update_fields_from_macro()
        np.testing.assert_allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], [ref.cl_smooth, ref.cd_smooth, ref.sep_frac], rtol=1e-12)
        assert np.array_equal(wt.render_rgba("vort"), ref.render_rgba("vort"))
        assert wt.stats().separation == ref.stats().separation


def test_rccl_plumbing_selftest(pkg):
    """ncclGetUniqueId / ncclCommInitRank / grouped ncclSend+ncclRecv from inside libwindtunnel on one GPU
    (one-rank communicator, messages to self): the calls the multi-GPU ghost exchange is made of."""
    pkg.Engine.comm_selftest(0, 4096)
    ident = pkg.Engine.comm_unique_id()
    assert len(ident) == 128 and any(ident)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_every_rank_plans_the_same_schedule(pkg, dtype):
    """Over RCCL each rank decides alone when to run a fused pass, how long, and when to refresh its ghost columns, and the exchange is
    collective: all ranks of a tunnel must decide alike, whatever their own width (edge slabs hold one ghost zone, interior slabs two,
    and a split that does not divide leaves widths one column apart).  wt_plan_steps returns those decisions without taking them."""
    import numpy as np
    cases = [(4096, 4096, 8, 16), (4096, 4096, 8, 17), (4096, 4096, 4, 16), (4096, 4096, 2, 9), (1984, 4096, 4, 16), (2304, 4096, 2, 2),
             (2304, 4096, 2, 1), (2304, 4096, 3, 3), (4000, 4096, 7, 16), (4099, 2048, 8, 5), (1100, 4096, 2, 16), (16384, 4096, 8, 16)]
    for nx, ny, nranks, halo in cases:
        if dtype == "float64" and nx * ny > 4096 * 4096:
            continue
        es = [pkg.Engine(nx, ny, dtype=dtype, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
        try:
            mask = np.zeros((ny, nx), np.uint8)
            for e in es:
                e.set_mask(mask); e.init_equilibrium(0.06)
            depths = {(e.get_option("fuse_active"), e.get_option("fuse_depth")) for e in es}
            assert len(depths) == 1, (nx, ny, nranks, halo, depths)
            for nsteps in (1, 4, 5, 17, 33, 100):
                plans = [tuple(e.plan_steps(nsteps, 0.58)) for e in es]
                assert len(set(plans)) == 1, (nx, ny, nranks, halo, nsteps, plans)
                p = plans[0]
                assert sum(abs(k) if k != -1 else 1 for k in p) == nsteps
                if halo >= 4 and nsteps > halo and es[0].get_option("fuse_active"):
                    assert p.count(-1) >= 1 and any(k >= 2 for k in p)            # refreshes AND fused passes (an initial state's ghosts last `halo` steps)
        finally:
            for e in es:
                e.close()


def test_the_planned_schedule_is_the_one_that_runs(pkg):
    """wt_plan_steps against what wt_step_group then does (passes and single steps counted by the library)."""
    import numpy as np
    nx, ny, nranks, halo = 2048, 512, 4, 16
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask
    es = [pkg.Engine(nx, ny, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_option("fuse_steps", 2)
            e.set_mask(mask); e.init_equilibrium(0.06)
        for nsteps in (50, 7, 33):
            plan = es[0].plan_steps(nsteps, 0.58)
            assert all(e.plan_steps(nsteps, 0.58) == plan for e in es)
            before = [(e.get_option("passes"), e.get_option("single_steps")) for e in es]
            pkg.Engine.step_group(es, nsteps, 0.58, 0.06)
            for e, (p0, s0) in zip(es, before):
                assert e.get_option("passes") - p0 == sum(1 for k in plan if k >= 2)
                assert e.get_option("single_steps") - s0 == sum(1 for k in plan if k in (1, -1))
    finally:
        for e in es:
            e.close()


def test_slabs_that_would_take_different_schedules_are_refused(pkg):
    """Rank agreement is a checked property: two locally linked slabs with deliberately different knob values get
    WT_ERR_STATE naming the field at their first stepping call — not a divergent plan (over RCCL: a hang).  The same table is all-reduced
    inside wt_comm_init_rank and at the first wt_step after a change on the RCCL transport (tests/_rccl_worker.py)."""
    nx, ny, nranks, halo = 2400, 1024, 2, 16
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask
    for knob, v0, v1, needle in (("fuse_depth", 4, 3, "fuse_depth"), ("fast_div", 1, 0, "fast_div"), ("chain", 1, 0, "chain"),
                                 ("fuse_steps", 2, 0, "fuse_steps"), ("plan_columns", 0, 300, "plan_columns")):
        es = [pkg.Engine(nx, ny, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
        try:
            pkg.Engine.link_local(es)
            es[0].set_option(knob, v0); es[1].set_option(knob, v1)
            for e in es:
                e.set_mask(mask); e.init_equilibrium(0.06)
            with pytest.raises(pkg.WTError) as ei:
                pkg.Engine.step_group(es, 20, 0.58, 0.06)
            assert ei.value.code == -5 and "disagree" in str(ei.value) and needle in str(ei.value), str(ei.value)
            assert all(e.info().steps_done == 0 for e in es)                      # refused before anything ran
            # a retry WITHOUT touching any option is refused the same way (the check used to be marked done before its verdict,
            # and the second call of a refused group ran unchecked)
            with pytest.raises(pkg.WTError) as ei2:
                pkg.Engine.step_group(es, 20, 0.58, 0.06)
            assert ei2.value.code == -5 and "disagree" in str(ei2.value) and needle in str(ei2.value), str(ei2.value)
            assert all(e.info().steps_done == 0 for e in es)
            es[1].set_option(knob, v0)                                            # put right: the next call checks again and runs
            pkg.Engine.step_group(es, 20, 0.58, 0.06)
            assert all(e.info().steps_done == 20 and e.get_option("agree_checks") >= 1 for e in es)
        finally:
            for e in es:
                e.close()
    # a halo that differs is caught when the group is linked, as before
    es = [pkg.Engine(nx, ny, rank=0, nranks=2, halo=16), pkg.Engine(nx, ny, rank=1, nranks=2, halo=8)]
    try:
        with pytest.raises(pkg.WTError):
            pkg.Engine.link_local(es)
    finally:
        for e in es:
            e.close()


def test_strongly_uneven_split_plans_one_schedule(pkg):
    """eligibility and steps per pass come from the SPLIT (widest slab below 4 GiB, narrowest slab's columns per unit), never from
    a rank's own width — a 3000-column slab next to a 100-column one plan alike and run bit-identically to the whole lattice."""
    nx, ny, halo = 3300, 1024, 16
    edges = [0, 100, 3100, 3300]
    mask = pkg.geometry.build_geometry(nx, ny, 5.0, None, "naca2412").mask
    f0, m0, _, _ = _single(pkg, mask, [40], 0.58, 0.06, "float32")
    es = [pkg.Engine(nx, ny, rank=r, nranks=3, halo=halo, edges=edges) for r in range(3)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.init_equilibrium(0.06)
        assert len({(e.get_option("fuse_active"), e.get_option("fuse_depth")) for e in es}) == 1
        for n in (1, 5, 17, 40, 100):
            assert len({tuple(e.plan_steps(n, 0.58)) for e in es}) == 1
        pkg.Engine.step_group(es, 40, 0.58, 0.06)
        f1 = np.concatenate([e.read_f() for e in es], axis=2)
        assert all(e.get_option("chain_downgrades") == 0 for e in es)
    finally:
        for e in es:
            e.close()
    assert bits_equal(f0, f1)


def test_exchange_timing_of_a_local_group(pkg):
    """option exchange_timing: events around every ghost exchange and the interior kernel beside it (bench.py --gpus N prints them per rank)."""
    nx, ny, nranks, halo = 2048, 1024, 2, 17
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask
    es = [pkg.Engine(nx, ny, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.init_equilibrium(0.06)
            e.set_option("exchange_timing", 1)
        pkg.Engine.step_group(es, 17 * 40, 0.58, 0.06)          # 40 refresh cycles: the ring of 32 is folded once on the way
        for e in es:
            n = e.get_option("exchanges")
            assert 38 <= n <= 41, n
            x, i, w = e.get_option("exchange_ms"), e.get_option("interior_ms"), e.get_option("exchange_exposed_ms")
            assert x > 0 and i > 0 and w >= 0 and x / n < 5.0 and i / n < 5.0, (x, i, w)
            e.set_option("exchange_timing", 0)
            assert e.get_option("exchanges") == 0
    finally:
        for e in es:
            e.close()


def test_pass_depth_reports_the_pass_actually_taken(pkg):
    """"fuse_depth" is what the tables are built for, "pass_depth" what a full pass takes for the tau in use — 3 on a four-step
    fp32 plan while the division by tau has to be the IEEE one."""
    nx, ny = 4096, 1024
    mask = pkg.geometry.build_geometry(nx, ny, 5.0, None, "naca2412").mask
    with pkg.Engine(nx, ny) as e:
        e.set_option("fuse_depth", 4); e.set_option("fuse_steps", 2)
        e.set_mask(mask); e.init_equilibrium(0.06)
        e.step(8, 0.58, 0.06)
        assert (e.get_option("fuse_depth"), e.get_option("pass_depth"), e.get_option("passes")) == (4, 4, 2)
        e.set_option("fast_div", 0)
        e.init_equilibrium(0.06)
        assert e.get_option("passes") == 0                    # counted since the last init / write_f, like single_steps
        e.step(9, 0.58, 0.06)
        assert (e.get_option("fuse_depth"), e.get_option("pass_depth"), e.get_option("passes")) == (4, 3, 3)


@pytest.mark.parametrize("dtype,halo", [("float32", 17), ("float32", 9), ("float64", 17)])
def test_trimmed_ghost_marching_changes_no_bit(pkg, dtype, halo):
    """trim_ghosts: between two refreshes a pass marches only the ghost columns that will still be exact after it (one unit list per remaining
    depth, cut lazily).  Owned columns, (rho, ux, uy) and the reductions equal the single lattice and the untrimmed group bit for bit through
    several refresh cycles, uneven step counts included."""
    nx, ny, nranks = 2304, 1024, 3
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask
    chunks = [3 * halo + 5, 7, 4 * halo, 9]
    f0, m0, r0, F0 = _single(pkg, mask, chunks, 0.58, 0.06, dtype)
    out = {}
    for trim in (1, 0):
        es = [pkg.Engine(nx, ny, dtype=dtype, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
        try:
            pkg.Engine.link_local(es)
            for e in es:
                e.set_option("fuse_steps", 2)
                e.set_option("trim_ghosts", trim)
                e.set_mask(mask); e.init_equilibrium(0.06)
            for n in chunks:
                pkg.Engine.step_group(es, n, 0.58, 0.06)
            out[trim] = (np.concatenate([e.read_f() for e in es], axis=2), [np.concatenate(p, axis=1) for p in zip(*[e.read_macro() for e in es])],
                         [e.get_option("trimmed_passes") for e in es], [e.get_option("passes") for e in es])
        finally:
            for e in es:
                e.close()
    assert all(t > 0 for t in out[1][2]) and all(t == 0 for t in out[0][2]), (out[1][2], out[0][2])
    assert out[1][3] == out[0][3]                                  # the same schedule either way
    for trim in (1, 0):
        assert bits_equal(f0, out[trim][0])
        assert all(bits_equal(a, b) for a, b in zip(m0, out[trim][1]))


@pytest.mark.parametrize("dtype,nranks,halo", [("float32", 4, 17), ("float32", 2, 29), ("float64", 3, 13)])
def test_all_fused_refresh_cycle_equals_single_lattice(pkg, dtype, nranks, halo):
    """option refresh = 1: the ghost columns are renewed by an exchange at a pass boundary and EVERY step of a cycle is a
    fused one — single_steps stays 0 through more than three refresh cycles, the planned schedule (-2 = exchange) is the one that runs, and
    owned columns and (rho, ux, uy) equal the single lattice bit for bit."""
    nx, ny = 600 * nranks, 1024
    mask = pkg.geometry.build_geometry(nx, ny, 7.0, None, "naca2412").mask
    chunks = [2 * halo, halo + 8, 40, 4 * halo]
    f0, m0, _, _ = _single(pkg, mask, chunks, 0.58, 0.06, dtype)
    es = [pkg.Engine(nx, ny, dtype=dtype, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_option("refresh", 1)
            e.set_mask(mask); e.init_equilibrium(0.06)
        assert all(e.get_option("fuse_active") == 1.0 for e in es)
        for n in chunks:
            plan = es[0].plan_steps(n, 0.58)
            assert all(e.plan_steps(n, 0.58) == plan for e in es)
            assert all(k >= 2 or k == -2 for k in plan) and sum(k for k in plan if k > 0) == n       # fused passes and boundary exchanges only
            x0 = [e.get_option("boundary_exchanges") for e in es]
            pkg.Engine.step_group(es, n, 0.58, 0.06)
            assert [e.get_option("boundary_exchanges") - a for e, a in zip(es, x0)] == [plan.count(-2)] * nranks
        assert all(e.get_option("single_steps") == 0 for e in es)
        assert all(e.get_option("boundary_exchanges") >= 4 for e in es)
        f1 = np.concatenate([e.read_f() for e in es], axis=2)
        m1 = [np.concatenate(p, axis=1) for p in zip(*[e.read_macro() for e in es])]
    finally:
        for e in es:
            e.close()
    assert bits_equal(f0, f1)
    assert all(bits_equal(a, b) for a, b in zip(m0, m1))
    # a group whose slabs chose different refresh modes is refused (the modes exchange at different steps)
    es = [pkg.Engine(nx, ny, dtype=dtype, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
    try:
        pkg.Engine.link_local(es)
        for r, e in enumerate(es):
            e.set_option("refresh", 1 if r == 0 else 0)
            e.set_mask(mask); e.init_equilibrium(0.06)
        with pytest.raises(pkg.WTError) as ei:
            pkg.Engine.step_group(es, 20, 0.58, 0.06)
        assert "disagree" in str(ei.value) and "refresh" in str(ei.value)
    finally:
        for e in es:
            e.close()


@pytest.mark.parametrize("dtype,nranks,halo,shape,aoa", [("float32", 4, 17, "naca2412", 7.0), ("float32", 2, 29, "naca6409", 10.0), ("float64", 3, 13, "naca2412", 7.0),
                                                         ("float32", 3, 28, "naca0012", 8.0)])
def test_fused_renewal_cycle_equals_single_lattice(pkg, dtype, nranks, halo, shape, aoa):
    """option refresh = 2: the ghost columns are renewed INSIDE a fused pass — the exchange runs beside the marching of the
    interior columns, the edge strips follow once the ghosts have landed.  Through more than three refresh cycles no step is a single k_step
    (`single_steps` 0), no exchange stands alone at a pass boundary (`boundary_exchanges` 0), the planned schedule (100 + k = a renewing pass of k
    steps) is the one that runs, and owned columns and (rho, ux, uy) equal the single lattice bit for bit (STEP_FS html:283-360, the split SURVEY 8e)."""
    nx, ny = 600 * nranks, 1024
    mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
    chunks = [2 * halo, halo + 8, 40, 4 * halo, 5]
    f0, m0, _, _ = _single(pkg, mask, chunks, 0.58, 0.06, dtype)
    es = [pkg.Engine(nx, ny, dtype=dtype, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_option("refresh", 2)
            e.set_mask(mask); e.init_equilibrium(0.06)
        assert all(e.get_option("fuse_active") == 1.0 and e.get_option("fuse_depth") >= 3 for e in es)
        renewals = 0
        for n in chunks:
            plan = es[0].plan_steps(n, 0.58)
            assert all(e.plan_steps(n, 0.58) == plan for e in es)
            assert all(k >= 2 for k in plan) and sum(k - 100 if k > 100 else k for k in plan) == n          # fused passes only, some of them renewing
            r0 = [e.get_option("fused_renewals") for e in es]
            pkg.Engine.step_group(es, n, 0.58, 0.06)
            assert [e.get_option("fused_renewals") - a for e, a in zip(es, r0)] == [sum(1 for k in plan if k > 100)] * nranks
            renewals += sum(1 for k in plan if k > 100)
        assert renewals >= 4
        assert all(e.get_option("single_steps") == 0 and e.get_option("boundary_exchanges") == 0 for e in es)
        f1 = np.concatenate([e.read_f() for e in es], axis=2)
        m1 = [np.concatenate(p, axis=1) for p in zip(*[e.read_macro() for e in es])]
    finally:
        for e in es:
            e.close()
    assert bits_equal(f0, f1)
    assert all(bits_equal(a, b) for a, b in zip(m0, m1))


def test_fused_renewal_after_write_f_and_with_a_mask_change(pkg):
    """refresh = 2 from a state whose ghosts are stale from the start (wt_write_f: the first thing a slab does is a renewing pass whose halo lines
    all come from the lattice) and across a mask upload (the AoA slider: the seam buffer is stale, the renewal plans are cut again)."""
    nx, ny, nranks, halo = 1800, 512, 3, 16
    geo = pkg.geometry
    m_a, m_b = geo.build_geometry(nx, ny, 4.0, None, "naca2412").mask, geo.build_geometry(nx, ny, 9.0, None, "naca2412").mask
    with pkg.Engine(nx, ny) as ref:
        ref.set_option("fuse_steps", 0)
        ref.set_mask(m_a); ref.init_equilibrium(0.06); ref.step(37, 0.58, 0.06)
        f_mid = ref.read_f()
        ref.step(50, 0.58, 0.06); ref.set_mask(m_b); ref.step(45, 0.58, 0.06)
        f_ref = ref.read_f()
    es = [pkg.Engine(nx, ny, rank=r, nranks=nranks, halo=halo) for r in range(nranks)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_option("refresh", 2); e.set_option("fuse_steps", 2)
            e.set_mask(m_a); e.init_equilibrium(0.06)
            e.write_f(np.ascontiguousarray(f_mid[:, :, e.x0:e.x0 + e.width]))
        pkg.Engine.step_group(es, 50, 0.58, 0.06)
        for e in es:
            e.set_mask(m_b)
        pkg.Engine.step_group(es, 45, 0.58, 0.06)
        assert all(e.get_option("single_steps") == 0 and e.get_option("fused_renewals") >= 5 for e in es)
        f1 = np.concatenate([e.read_f() for e in es], axis=2)
    finally:
        for e in es:
            e.close()
    assert bits_equal(f_ref, f1)
