"""N>1 path on CPU: world_size-2 and -3 gloo runs of the slab host logic
(airfoil_cfd_tool_amd.distributed.SlabWindTunnel) with a stand-in engine (tests/_slab_standin.py:
the oracle per slab + gloo send/recv following libwindtunnel's deep-halo protocol).  Checks that the
sharded tunnel equals the monolithic oracle bit for bit and that reductions combine correctly."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,halo", [(2, 1), (2, 4), (3, 2)])
def test_slab_tunnel_over_gloo(world, halo):
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(r), str(world), str(port), str(halo)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("gloo worker timed out")
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
    assert "rank0 checks PASS" in outs[0]


def test_slab_bounds(pkg):
    from airfoil_cfd_tool_amd.distributed import slab_bounds
    assert slab_bounds(4096, 8) == [(512 * r, 512) for r in range(8)]
    assert slab_bounds(10, 3) == [(0, 3), (3, 3), (6, 4)]
    b = slab_bounds(16384, 8)
    assert b[0] == (0, 2048) and b[-1] == (14336, 2048)


def test_slabs_cut_by_cost(pkg):
    """distributed.balanced_edges / balance_split: host logic of the split by measured cost (no GPU: the 'measurement' is a model)."""
    from airfoil_cfd_tool_amd.distributed import balance_split, balanced_edges, slab_bounds, slab_edges
    assert slab_edges(4096, 8) == [512 * r for r in range(9)]
    assert slab_bounds(100, 3, [0, 10, 50, 100]) == [(0, 10), (10, 40), (50, 50)]
    assert balanced_edges([0, 10, 20], [1.0, 1.0], 3) == [0, 10, 20]
    e = balanced_edges([0, 10, 20], [1.0, 3.0], 3)
    assert e[0] == 0 and e[2] == 20 and 10 < e[1] <= 17                     # the dear slab gets narrower
    assert balanced_edges([0, 10, 20], [0.0, 3.0], 8) == [0, 12, 20]         # minimum width kept
    with pytest.raises(ValueError):
        balanced_edges([0, 10, 20], [1.0, 1.0], 11)

    # a tunnel whose columns 1000..3000 cost 2.5 x the plain ones, plus a fixed cost per slab
    dens = np.ones(4096); dens[1000:3000] = 2.5

    def measure(edges):
        return [5.0 + 0.02 * dens[a:b].sum() for a, b in zip(edges[:-1], edges[1:])]
    best, hist = balance_split(4096, 8, 32, measure, rounds=4)
    assert best[0] == 0 and best[-1] == 4096 and all(b - a >= 32 for a, b in zip(best[:-1], best[1:]))
    assert hist[0][0] == slab_edges(4096, 8)
    first, kept = max(hist[0][1]), max(measure(best))
    ideal = 5.0 + 0.02 * dens.sum() / 8
    assert kept < first and kept < 1.03 * ideal                             # equal widths: 30.6, ideal 22.7
    # a split that cannot be improved stays
    best2, hist2 = balance_split(4096, 8, 32, lambda ed: [1.0] * 8, rounds=3)
    assert best2 == slab_edges(4096, 8) and len(hist2) == 1
