"""GPU, ONE device: the library's multi-rank code as real processes, with a stand-in for RCCL underneath (tests/_rccl_stub/rccl_stub.cpp,
LD_PRELOADed in front of the PRODUCTION libwindtunnel.so).  RCCL itself refuses two ranks on one GPU ("Duplicate GPU detected"), and no box with
two GPUs has been available to any round — so until one is, this is where the code that only runs with nranks > 1 over TR_RCCL is executed at
all: wt_comm_init_rank with its schedule-agreement all-reduce (and the refusal of a deviating rank on every rank), the grouped ghost-column
exchange driven by every rank's own schedule, equal and caller-cut slabs, fp32 / fp64, two- and four-step plans, the worker's plan-agreement
check, and bench.py --gpus N end to end with its per-rank report.  The stand-in keeps NCCL's matching rules (in-order per pair, grouped calls
deadlock-free) and turns a size mismatch or a missing peer into an error instead of a hang; it says nothing about RCCL or xGMI themselves."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rccl_stub") / "librccl_stub.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "_rccl_stub", "rccl_stub.cpp"), "-ldl", "-lrt", "-lpthread"],
                   check=True, capture_output=True, timeout=300)
    return so


def _env(stub, **extra):
    env = dict(os.environ, LD_PRELOAD=stub, HSA_ENABLE_IPC_MODE_LEGACY="0", WT_RCCL_SAME_DEVICE="0")
    env.update(extra)
    return env


@pytest.mark.parametrize("world,halo", [(2, 16), (3, 5), (4, 29)])
def test_slab_ranks_over_the_stand_in_transport_equal_single_lattice(stub, world, halo):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + world + halo), os.path.join(HERE, "_rccl_worker.py"), str(halo)]
    r = subprocess.run(cmd, env=_env(stub), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    assert r.stdout.count("PASS") == 7 and "FAIL" not in r.stdout, r.stdout[-4000:]


def test_bench_two_ranks_end_to_end_over_the_stand_in_transport(stub):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank): slabs cut by measured cost, the library's
    communicator, warm-up, timed steps, the JSON line with the first-contact report of every rank."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29741",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "2048", "--ny", "1024", "--steps", "87", "--warmup", "29", "--cpu-steps", "0", "--balance", "1",
           "--halo", "29"]
    r = subprocess.run(cmd, env=_env(stub, WT_BENCH_FORCE_DEVICE="0", WT_BENCH_TORCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and len(d["device_ms"]) == 2
    assert d["comm_ranks_seen"] == [2] and [x["rank"] for x in d["ranks"]] == [0, 1]
    assert sum(d["config"]["slab_widths"]) == 2048 and d["config"]["halo"] == 29 and len(d["config"]["balance"]) >= 1
    for x in d["ranks"]:
        assert x["comm_ranks"] == 2 and x["exchanges"] >= 3 and x["exchange_ms_each"] > 0 and x["agree_checks"] >= 2
        assert x["fuse_active"] == 1 and x["passes"] > 0 and x["chain_downgrades"] == 0
    assert len({(x["fuse_depth"], x["pass_depth"]) for x in d["ranks"]}) == 1


def test_bench_two_ranks_without_a_launcher_over_the_stand_in_transport(stub):
    """`python bench.py --gpus 2` launched exactly like the N = 1 line (no torch.distributed.run in front, WORLD_SIZE unset): bench.py starts the
    two ranks itself as a child job before it touches the GPU and the N = 2 line comes out, exchange-cost model included."""
    env = {k: v for k, v in _env(stub, WT_BENCH_FORCE_DEVICE="0", WT_BENCH_TORCH_BACKEND="gloo").items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "WT_BENCH_SELF_LAUNCHED")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "2048", "--ny", "1024", "--steps", "58", "--warmup", "29", "--cpu-steps", "0",
           "--balance", "0", "--halo", "29"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "starting 2 ranks as a child job" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["comm_ranks_seen"] == [2] and [x["rank"] for x in d["ranks"]] == [0, 1]
    for x in d["ranks"]:
        # the exchange-cost model of the line: 9 populations x halo columns x pitch x element size per side and refresh
        assert x["exchange_bytes_each"] == 9 * 29 * 1024 * 4 and x["exchange_model_us"] > 0 and x["exchange_sides"] == 1


def test_bench_self_launch_on_one_gpu_fails_cleanly():
    """The same launcher-less command WITHOUT the stand-in on this one-GPU box: rank 1 finds no second device -> its one-line diagnosis, a non-zero
    status relayed by the parent, no bench line, nothing retried."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs: the command runs to the end")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "WT_BENCH_SELF_LAUNCHED", "LD_PRELOAD")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "1024", "--ny", "512", "--steps", "8", "--warmup", "4", "--cpu-steps", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0
    assert "[bench.py] rank 1 (device 1) FAILED" in r.stderr and "the 2-rank job exited with status" in r.stderr
    assert r.stderr.count("starting 2 ranks as a child job") == 1
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("{")] == []


@pytest.mark.parametrize("world,dtype", [(3, "float32"), (2, "float64")])
def test_slab_wind_tunnel_host_class_over_the_stand_in_transport(stub, world, dtype):
    """distributed.SlabWindTunnel on real slab engines: frame loop, AoA slider, combined reductions, gathered read-backs, and the vorticity field
    whose ghost columns move over the transport (refresh_macro_ghosts, TR_RCCL branch: never run before) — against one WindTunnel."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29780 + world), os.path.join(HERE, "_slab_tunnel_worker.py"), dtype]
    r = subprocess.run(cmd, env=_env(stub), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "PASS" in r.stdout and "FAIL" not in r.stdout, r.stdout[-4000:]
