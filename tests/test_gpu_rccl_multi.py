"""GPU, two or more devices: the RCCL transport of the column slabs (wt_comm_init_rank, exchange over ncclSend/ncclRecv on the
comm stream) across real ranks — one process per GPU under torch.distributed.run — against the single lattice, bit for bit.
Skipped on boxes with one GPU (the in-process transport of tests/test_gpu_slabs.py covers the slab logic there)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _device_count():
    import torch
    return torch.cuda.device_count()          # does not initialise the GPU in this process


@pytest.mark.parametrize("world,halo", [(2, 16), (2, 3), (4, 16)])
def test_rccl_slabs_equal_single_lattice(world, halo):
    if _device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29650 + world + halo), os.path.join(HERE, "_rccl_worker.py"), str(halo)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("PASS") == 7 and "FAIL" not in r.stdout, r.stdout[-3000:]
