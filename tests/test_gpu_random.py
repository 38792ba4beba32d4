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
    for fuse, depth in ((0, 0), (4, 2), (2, 2), (2, 3), (2, 4)):     # single steps; marching kernel: sites per lane, steps per pass
        with pkg.Engine(nx, ny, dtype=dtype) as e:
            e.set_option("fuse_steps", 0)
            if fuse:
                sites = fuse if not (dtype == "float64" and depth >= 3) else 0      # fp64, three steps: one site per lane, implied
                if (sites and ny % sites) or nx < (16 if depth >= 3 else 8) or (dtype == "float64" and fuse == 4):
                    continue
                e.set_option("fuse_chunk", int(rng.integers(1, 40)))
                if sites:
                    e.set_option("fuse_sites", sites)
                e.set_option("fuse_depth", depth)
                e.set_option("fuse_steps", 2)
            e.set_mask(mask); e.init_equilibrium(u0)
            for n in steps:
                e.step(n, tau, u0)
            f, m = e.read_f(), e.read_macro()
        assert bits_equal(f, ref_f), (seed, fuse, depth, nx, ny, dtype)
        assert all(bits_equal(a, b) for a, b in zip(m, ref_m)), (seed, fuse)
