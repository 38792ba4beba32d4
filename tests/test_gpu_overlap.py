"""GPU: the marching kernels on OVERLAPPING windows (option window_overlap; csrc/step_chain.hpp k_march3: windows of 128 rows that own the 120 in the
middle and carry four margin rows on either side instead of reading halo lines) — the same bits as single steps (STEP_FS, html:283-360, once per
step) on lattices whose height is and is not a multiple of the window stride, with the body in the first window, in the last, across a seam; the
automatic choice (lattices — whole ones and slabs alike — of up to 7.5 M local sites: on; larger ones: off); the options that exclude it."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu

CASES = [(320, 256, "naca0012", 4.0, 37), (768, 1000, "naca4412", 12.0, 29), (512, 120, "naca0012", 0.0, 23), (544, 1366, "naca6409", 10.0, 31),
         (400, 242, "naca2412", -6.0, 26), (4096, 16, "naca0012", 2.0, 19), (3000, 30, "naca2412", 5.0, 21), (2200, 122, "naca4412", 9.0, 17)]


def _run(pkg, nx, ny, mask, nsteps, opts, dtype="float32", tau=0.58):
    with pkg.Engine(nx, ny, dtype=dtype) as e:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_mask(mask)
        e.init_equilibrium(0.06)
        e.step(nsteps, tau, 0.06)
        return e.read_f(), e.read_macro(), {k: e.get_option(k) for k in ("window_overlap", "fuse_depth", "fuse_active", "single_steps")}


@pytest.mark.parametrize("depth", [4, 3])
@pytest.mark.parametrize("nx,ny,shape,aoa,nsteps", CASES, ids=[f"{c[0]}x{c[1]}" for c in CASES])
def test_overlapping_windows_equal_single_steps(pkg, nx, ny, shape, aoa, nsteps, depth):
    mask = pkg.geometry.build_geometry(nx, ny, aoa, None, shape).mask
    ref_f, ref_m, _ = _run(pkg, nx, ny, mask, nsteps, {"fuse_steps": 0})
    f, m, info = _run(pkg, nx, ny, mask, nsteps, {"fuse_steps": 2, "fuse_depth": depth, "window_overlap": 1})
    assert info["window_overlap"] == 1.0 and info["fuse_active"] == 1.0 and info["fuse_depth"] == depth
    assert bits_equal(f, ref_f)
    for a, b in zip(m, ref_m):
        assert bits_equal(a, b)


def test_overlap_is_automatic_for_slabs_and_small_lattices_and_never_for_fp64_or_fast_math(pkg):
    nx, ny = 640, 512
    mask = pkg.geometry.build_geometry(nx, ny, 5.0, None, "naca2412").mask
    with pkg.Engine(nx, ny) as e:                      # a whole lattice of up to 7.5 M sites: overlapping (cache-resident, not bound by line traffic)
        e.set_option("fuse_steps", 2)
        e.set_mask(mask)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("window_overlap") == 1.0
        e.set_option("window_overlap", 0)
        assert e.get_option("window_overlap") == 0.0
        e.set_option("window_overlap", -1)
        assert e.get_option("window_overlap") == 1.0
        e.set_option("fast_math", 1)                   # the contracted kernels know no overlapping windows: the plan is cut again
        assert e.get_option("window_overlap") == 0.0
        e.init_equilibrium(0.06)
        e.step(9, 0.58, 0.06)
        e.set_option("fast_math", 0)
        assert e.get_option("window_overlap") == 1.0
        with pytest.raises(pkg.WTError):
            e.set_option("window_overlap", 2)
    big = np.zeros((2048, 4096), np.uint8)
    big[1000:1040, 1000:1400] = 1
    with pkg.Engine(4096, 2048) as e:                  # a large lattice (8.4 M sites): windows that tile the column, whatever it plans like
        e.set_mask(big)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("window_overlap") == 0.0
        e.set_option("plan_columns", 541)
        assert e.get_option("window_overlap") == 0.0
        e.set_option("window_overlap", 1)
        assert e.get_option("window_overlap") == 1.0
    with pkg.Engine(nx, ny, dtype="float64") as e:
        e.set_option("fuse_steps", 2)
        e.set_option("window_overlap", 1)
        e.set_mask(mask)
        assert e.get_option("fuse_active") == 1.0 and e.get_option("window_overlap") == 0.0
    es = [pkg.Engine(nx, ny, rank=r, nranks=2, halo=9) for r in range(2)]          # the slabs of a split: overlapping
    try:
        pkg.Engine.link_local(es)
        for s in es:
            s.set_option("fuse_steps", 2)
            s.set_mask(mask)
        assert all(s.get_option("fuse_active") == 1.0 and s.get_option("window_overlap") == 1.0 for s in es)      # (by their local size, like whole lattices)
    finally:
        for s in es:
            s.close()


@pytest.mark.parametrize("overlap,refresh", [(0, 0), (1, 0), (0, 2), (1, 2), (0, 1), (1, 1)])
def test_slab_group_equals_single_lattice_on_either_window_layout(pkg, overlap, refresh):
    """Three local slabs of a tunnel with the body across two of them, through ghost renewals of every kind (refresh 0 / 1 / 2) and trimmed ghost
    passes, on overlapping windows (the automatic choice for slabs) and on windows that tile the column (the layout of whole lattices, forced): owned
    columns bit-identical to the single lattice."""
    nx, ny, nsteps = 900, 744, 47
    mask = pkg.geometry.build_geometry(nx, ny, 8.0, None, "naca4412").mask
    ref_f, ref_m, _ = _run(pkg, nx, ny, mask, nsteps, {"fuse_steps": 0})
    es = [pkg.Engine(nx, ny, rank=r, nranks=3, halo=13) for r in range(3)]
    try:
        pkg.Engine.link_local(es)
        for s in es:
            s.set_option("fuse_steps", 2)
            s.set_option("window_overlap", overlap)
            s.set_option("refresh", refresh)
            s.set_mask(mask)
            s.init_equilibrium(0.06)
        for n in (30, 17):
            pkg.Engine.step_group(es, n, 0.58, 0.06)
        assert all(s.get_option("window_overlap") == float(overlap) and s.get_option("fuse_active") == 1.0 for s in es)
        assert sum(s.get_option("passes") for s in es) > 0
        assert bits_equal(np.concatenate([s.read_f() for s in es], axis=2), ref_f)
        for k, b in enumerate(ref_m):
            assert bits_equal(np.concatenate([s.read_macro()[k] for s in es], axis=1), b)
    finally:
        for s in es:
            s.close()
