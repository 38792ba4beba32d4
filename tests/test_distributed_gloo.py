"""N>1 path on CPU: world_size-2 and -3 gloo runs of the slab host logic
(airfoil_cfd_tool_amd.distributed.SlabWindTunnel) with a stand-in engine (tests/_slab_standin.py:
the oracle per slab + gloo send/recv following libwindtunnel's deep-halo protocol).  Checks that the
sharded tunnel equals the monolithic oracle bit for bit and that reductions combine correctly."""
import os
import socket
import subprocess
import sys

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
