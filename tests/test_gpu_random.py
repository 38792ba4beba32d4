"""GPU: seeded random lattices, masks and parameters against the C oracle, with and without the
two-steps-per-launch mode.  Random speckle/blocks masks hit tile-class and window-seam corner cases
that airfoil shapes do not."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def _random_mask(rng, nx, ny):
    m = np.zeros((ny, nx), np.uint8)
    kind = rng.integers(0, 4)
    if kind == 0:                                    # sparse speckles
        m[rng.random((ny, nx)) < 0.002] = 1
    elif kind == 1:                                  # a few rectangles, some touching the borders
        for _ in range(rng.integers(1, 6)):
            x0, y0 = rng.integers(0, nx), rng.integers(0, ny)
            m[y0:y0 + rng.integers(1, ny // 3 + 2), x0:x0 + rng.integers(1, nx // 3 + 2)] = 1
    elif kind == 2:                                  # thin lines crossing tile / window seams
        for r in (251, 252, 253, 254, 255, 256, 257, 503, 504, 505):
            if r < ny:
                x0 = rng.integers(0, max(1, nx - 40))
                m[r, x0:x0 + rng.integers(1, 40)] = 1
        m[:, rng.integers(3, nx - 3)] = rng.random(ny) < 0.3
    return m                                          # kind 3: empty


@pytest.mark.parametrize("seed", range(12))
def test_random_case(pkg, oracle_c, seed):
    rng = np.random.default_rng(1000 + seed)
    nx = int(rng.integers(16, 700))
    ny = int(rng.choice([64, 128, 252, 256, 260, 300, 508, 512, 516, 768, 1024])) if seed % 2 else int(rng.integers(16, 600))
    dtype = "float64" if seed % 5 == 4 else "float32"
    tau = float(rng.uniform(0.51, 1.2))
    u0 = float(rng.uniform(0.02, 0.11))
    steps = [int(v) for v in rng.integers(1, 14, size=3)]
    mask = _random_mask(rng, nx, ny)
    ref_f, ref_m = oracle_c.run(mask, sum(steps), tau, u0, np.dtype(dtype))
    for fuse, depth in ((0, 0), (2, 2), (2, 3), (2, 4)):     # single steps; marching kernels: steps per pass
        with pkg.Engine(nx, ny, dtype=dtype) as e:
            e.set_option("fuse_steps", 0)
            if fuse:
                sites = 2 if dtype == "float32" else 1
                if ny % sites or nx < (8 if depth == 2 and dtype == "float32" else 16):
                    continue
                e.set_option("fuse_chunk", int(rng.integers(1, 40)))
                e.set_option("fuse_depth", depth)
                e.set_option("fuse_steps", 2)
            e.set_mask(mask); e.init_equilibrium(u0)
            for n in steps:
                e.step(n, tau, u0)
            f, m = e.read_f(), e.read_macro()
        assert bits_equal(f, ref_f), (seed, fuse, depth, nx, ny, dtype)
        assert all(bits_equal(a, b) for a, b in zip(m, ref_m)), (seed, fuse)


@pytest.mark.parametrize("dtype,nx,ny", [("float32", 1500, 1280), ("float64", 1300, 768)])
def test_random_large_default_plan(pkg, oracle_c, dtype, nx, ny):
    """A lattice large enough for the AUTOMATIC plan (several rounds of long units, many windows), speckles + blocks + lines on the
    seams of 64-, 128- and 256-row windows and on both tunnel ends; default options, then every depth forced; uneven step counts."""
    rng = np.random.default_rng(4242)
    m = np.zeros((ny, nx), np.uint8)
    m[rng.random((ny, nx)) < 0.0015] = 1
    for r in (63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 511, 512, 513):
        if r < ny:
            x0 = int(rng.integers(0, nx - 60))
            m[r, x0:x0 + int(rng.integers(1, 60))] = 1
    for _ in range(5):
        x0, y0 = int(rng.integers(0, nx)), int(rng.integers(0, ny))
        m[y0:y0 + int(rng.integers(2, 90)), x0:x0 + int(rng.integers(2, 120))] = 1
    m[ny // 3: ny // 3 + 40, 0:3] = 1; m[ny // 2: ny // 2 + 30, nx - 3:nx] = 1; m[0, 100:140] = 1; m[ny - 1, 300:350] = 1
    steps = [4, 3, 7, 1, 6]
    tau, u0 = 0.56, 0.07
    ref_f, ref_m = oracle_c.run(m, sum(steps), tau, u0, np.dtype(dtype))
    for depth in (0, 2, 3, 4):
        with pkg.Engine(nx, ny, dtype=dtype) as e:
            if depth:
                e.set_option("fuse_depth", depth)
                e.set_option("fuse_steps", 2)
            e.set_mask(m); e.init_equilibrium(u0)
            assert depth == 0 or e.get_option("fuse_active") == 1.0
            for n in steps:
                e.step(n, tau, u0)
            f, mac = e.read_f(), e.read_macro()
        assert bits_equal(f, ref_f), (dtype, depth)
        assert all(bits_equal(a, b) for a, b in zip(mac, ref_m)), (dtype, depth)
