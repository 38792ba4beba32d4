"""GPU parity on BASELINE.json's configurations at their FULL lattice sizes.

configs[0] (256x128, 500 steps) is covered by the reference-shader golden
(tests/test_gpu_golden_and_reductions.py).  Here: configs[1] in full (1024x512 fp32, NACA 2412, 5 deg,
2000 steps) and configs[2..4] at full size for a bounded number of steps, compared bit for bit with
the C oracle, plus size-independent properties (mass drift, symmetry, far-field recovery, slab
decomposition == single lattice)."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def _gpu(pkg, mask, steps, tau, u0, dtype):
    ny, nx = mask.shape
    with pkg.Engine(nx, ny, dtype=dtype) as e:
        e.set_mask(mask); e.init_equilibrium(u0); e.step(steps, tau, u0)
        return e.read_f(), e.read_macro()


def test_config1_naca2412_1024x512_2000_steps(pkg, oracle_c, oracle_np):
    mask = pkg.geometry.build_geometry(1024, 512, 5.0, None, "naca2412").mask
    assert int((mask != 0).sum()) == 25283
    f, m = _gpu(pkg, mask, 2000, 0.58, 0.06, "float32")
    fr, mr = oracle_c.run(mask, 2000, 0.58, 0.06, np.float32)
    assert np.abs(m[0] - mr[0]).max() <= 1e-5 and max(np.abs(m[1] - mr[1]).max(), np.abs(m[2] - mr[2]).max()) <= 5e-6
    assert bits_equal(f, fr) and all(bits_equal(a, b) for a, b in zip(m, mr))
    # physics sanity: lift is positive at +5 deg, the flow is attached
    fx, fy, surf, rev = oracle_np.compute_forces_raw(m[0], m[1], mask)
    assert fy > 0 and fx > 0 and rev / surf < 0.25


def test_config2_4096x4096_fp32_full_size(pkg, oracle_c):
    mask = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
    f, m = _gpu(pkg, mask, 24, 0.58, 0.06, "float32")
    fr, mr = oracle_c.run(mask, 24, 0.58, 0.06, np.float32)
    assert bits_equal(f, fr) and all(bits_equal(a, b) for a, b in zip(m, mr))


def test_config4_4096x2048_fp64_re1e6_full_size(pkg, oracle_c):
    tau = pkg.tau_from_reynolds(1e6, 0.06, 4096)
    mask = pkg.geometry.build_geometry(4096, 2048, 12.0, None, "naca4412").mask
    assert int((mask != 0).sum()) == 405515
    f, m = _gpu(pkg, mask, 24, tau, 0.06, "float64")
    fr, mr = oracle_c.run(mask, 24, tau, 0.06, np.float64)
    assert bits_equal(f, fr) and all(bits_equal(a, b) for a, b in zip(m, mr))


def test_config3_16384x4096_slabs_vs_single_and_oracle(pkg, oracle_c):
    """configs[3]'s lattice: 8 column slabs (in-process transport) == single lattice == oracle."""
    nx, ny, steps = 16384, 4096, 6
    mask = pkg.geometry.build_geometry(nx, ny, 8.0, None, "naca0012").mask
    f, m = _gpu(pkg, mask, steps, 0.58, 0.06, "float32")
    fr, mr = oracle_c.run(mask, steps, 0.58, 0.06, np.float32)
    assert bits_equal(f, fr) and all(bits_equal(a, b) for a, b in zip(m, mr))
    del fr, mr
    es = [pkg.Engine(nx, ny, rank=r, nranks=8, halo=4) for r in range(8)]
    try:
        pkg.Engine.link_local(es)
        for e in es:
            e.set_mask(mask); e.init_equilibrium(0.06)
        pkg.Engine.step_group(es, steps, 0.58, 0.06)
        for e in es:
            assert bits_equal(e.read_f(), np.ascontiguousarray(f[:, :, e.x0:e.x0 + e.width]))
    finally:
        for e in es:
            e.close()


def test_config3_16384x4096_slabs_through_two_refresh_cycles(pkg):
    """configs[3] at full size, halo 16, 2*halo + 2 steps: every one of the 8 in-process slabs refreshes its ghost
    columns twice (exchange on the second stream beside the interior columns, edge strips behind the event) and the
    result equals the single lattice bit for bit — single-step kernels on the slabs, then the marching kernel too."""
    nx, ny, halo = 16384, 4096, 16
    steps = 2 * halo + 2
    mask = pkg.geometry.build_geometry(nx, ny, 8.0, None, "naca0012").mask
    f, m = _gpu(pkg, mask, steps, 0.58, 0.06, "float32")
    for fuse in (0, 2):
        es = [pkg.Engine(nx, ny, rank=r, nranks=8, halo=halo) for r in range(8)]
        try:
            pkg.Engine.link_local(es)
            for e in es:
                e.set_option("fuse_steps", fuse)
                e.set_mask(mask); e.init_equilibrium(0.06)
            if fuse:
                assert all(e.get_option("fuse_active") == 1.0 for e in es)
            pkg.Engine.step_group(es, steps, 0.58, 0.06)
            for e in es:
                assert e.info().steps_done == steps
                assert bits_equal(e.read_f(), np.ascontiguousarray(f[:, :, e.x0:e.x0 + e.width]))
                assert all(bits_equal(a, np.ascontiguousarray(b[:, e.x0:e.x0 + e.width])) for a, b in zip(e.read_macro(), m))
        finally:
            for e in es:
                e.close()


def test_config4_full_size_stall_label_and_vorticity_sign(pkg, oracle_c, oracle_np):
    """configs[4] at FULL size (4096x2048 fp64, NACA 4412, 12 deg, Re 1e6 -> tau ~ 0.5004): 20 frames of 12 steps with
    the force read-out after each (html:650-700), then the stall label (html:862-885) and the sign structure of the
    vorticity field (html:411-417) against the C oracle run in lock step; populations bit-identical at the end."""
    nx, ny = 4096, 2048
    tau = pkg.tau_from_reynolds(1e6, 0.06, nx)
    with pkg.WindTunnel(shape="naca4412", nx=nx, ny=ny, aoa_deg=12.0, tau=tau, dtype="float64") as wt:
        assert int((wt.geometry.mask != 0).sum()) == 405515
        st = oracle_np.ForceState()
        f = None
        for _ in range(20):
            wt.sim_step(12)
            wt.compute_forces()
            f, mac = oracle_c.run(wt.geometry.mask, 12, tau, 0.06, np.float64, f=f)
            st.update(*oracle_np.compute_forces_raw(mac[0], mac[1], wt.geometry.mask), 0.06, nx)
        assert wt.stats().separation == oracle_np.stall_label(st.sep)
        np.testing.assert_allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], [st.cl, st.cd, st.sep], rtol=1e-9, atol=1e-12)
        assert bits_equal(wt.read_f(), f)
        wt.update_fields_from_macro()
        t = wt.render_field(field="vort")
        ref = oracle_np.field_scalar(2, *mac, wt.geometry.mask, 0.06, wt.max_s, wt.cp_min, wt.cp_max)
        assert np.array_equal(np.sign(np.nan_to_num(t)), np.sign(np.nan_to_num(ref)))
        assert np.array_equal(np.isnan(t), wt.geometry.mask != 0)


def test_low_tau_fp64_stall_indicator_matches_oracle(pkg, oracle_c, oracle_np):
    """configs[4] in miniature (fp64, Re-derived tau ~0.5004, 12 deg): separation label and
    vorticity sign structure — identical because the fields are bit-identical."""
    nx, ny = 1024, 512
    tau = pkg.tau_from_reynolds(1e6 * nx / 4096, 0.06, nx)
    with pkg.WindTunnel(shape="naca4412", nx=nx, ny=ny, aoa_deg=12.0, tau=tau, dtype="float64") as wt:
        st = oracle_np.ForceState()
        f = None
        for _ in range(30):
            wt.sim_step(12)
            wt.compute_forces()
            f, mac = oracle_c.run(wt.geometry.mask, 12, tau, 0.06, np.float64, f=f)
            st.update(*oracle_np.compute_forces_raw(mac[0], mac[1], wt.geometry.mask), 0.06, nx)
        assert wt.stats().separation == oracle_np.stall_label(st.sep)
        np.testing.assert_allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], [st.cl, st.cd, st.sep], rtol=1e-9, atol=1e-12)
        wt.update_fields_from_macro()
        t = wt.render_field(field="vort")
        ref = oracle_np.field_scalar(2, *mac, wt.geometry.mask, 0.06, wt.max_s, wt.cp_min, wt.cp_max)
        assert np.array_equal(np.sign(np.nan_to_num(t)), np.sign(np.nan_to_num(ref)))


def test_symmetric_airfoil_zero_lift_and_mass_drift(pkg):
    """Size-independent properties on 2048x1024: NACA 0012 at 0 deg keeps CL ~ 0 (mirror symmetry of
    scheme and mask rows); total mass drifts only through the open boundaries."""
    nx, ny = 2048, 1024
    with pkg.WindTunnel(shape="naca0012", nx=nx, ny=ny, aoa_deg=0.0) as wt:
        wt.sim_step(300)
        rho, ux, uy = wt.read_macro()
        fluid = wt.geometry.mask == 0
        wt.compute_forces()
        assert abs(wt.cl_smooth) < 5e-3 * max(1.0, abs(wt.cd_smooth))
        assert abs(float(rho[fluid].astype(np.float64).mean()) - 1.0) < 2e-3
        # far field recovers the free stream
        assert np.abs(ux[:, :8] - np.float32(0.06)).max() < 2e-3 and np.abs(uy[:, :8]).max() < 2e-3
        if np.array_equal(wt.geometry.mask, wt.geometry.mask[::-1]):
            assert np.abs(ux - ux[::-1]).max() < 1e-5 and np.abs(uy + uy[::-1]).max() < 1e-5
