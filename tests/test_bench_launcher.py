"""bench.py's N > 1 entry: it must fail CLEANLY and diagnosably — nothing here needs more than the
machine it runs on.

* without a launcher `--gpus 2` starts the two ranks itself, as a child `python -m torch.distributed.run ...` job, BEFORE it imports torch or touches
  the GPU, relays the child's output and exit status.  CPU box: no HIP device -> both ranks fail with their one-line diagnosis,
  the job's status is non-zero, no JSON line; 1-GPU box: rank 1 fails that way; over the RCCL stand-in (tests/test_gpu_rccl_stub.py) the same
  launcher-less command prints the N = 2 line;
* (GPU) under torch.distributed.run with more ranks than visible GPUs every surplus rank prints one line with its rank,
  device and the reason and exits non-zero, the launcher tears the job down, no JSON line is printed; on a box that DOES
  have two GPUs the same command must instead print the JSON line with per-rank device times.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _no_launcher_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "WT_BENCH_SELF_LAUNCHED")}
    env.update(extra)
    return env


def test_gpus_without_launcher_self_launches_and_fails_cleanly_without_devices():
    """No GPU here: the self-launched ranks cannot find a HIP device -> one diagnosable line per rank, non-zero status, no bench line, no retry."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a box with two GPUs runs this command to the end (tests/test_gpu_rccl_multi.py)")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--nx", "256", "--ny", "128", "--steps", "4", "--warmup", "2", "--cpu-steps", "0"],
                       capture_output=True, text=True, timeout=600, env=_no_launcher_env())
    assert p.returncode != 0
    assert "starting 2 ranks as a child job" in p.stderr and "torch.distributed.run" in p.stderr and "--nproc-per-node 2" in p.stderr
    assert "[bench.py] rank 1 (device 1) FAILED" in p.stderr and "HIP device(s) are visible" in p.stderr
    assert "the 2-rank job exited with status" in p.stderr
    assert [ln for ln in p.stdout.splitlines() if ln.startswith("{")] == []
    assert p.stderr.count("starting 2 ranks as a child job") == 1               # launched once: nothing is retried


def test_self_launch_happens_before_torch_is_imported():
    """The parent of a self-launched job must never have initialised the GPU: it imports neither torch nor the library before it starts the child."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2'];\n"
            "import subprocess\n"
            "def fake_run(cmd, env=None):\n"
            "    assert 'torch' not in sys.modules and 'airfoil_cfd_tool_amd' not in sys.modules, 'imported before the launch'\n"
            "    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and cmd[-2:] == ['--gpus', '2'] and '--nproc-per-node' in cmd\n"
            "    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and env.get('WT_BENCH_SELF_LAUNCHED') == '1'\n"
            "    class R: returncode = 7\n"
            "    return R()\n"
            "subprocess.run = fake_run\n"
            f"runpy.run_path({BENCH!r}, run_name='__main__')\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=_no_launcher_env())
    assert p.returncode == 7, (p.stdout, p.stderr)                              # the child's status is the parent's
    assert "exited with status 7" in p.stderr


@pytest.mark.gpu
def test_two_ranks_fail_cleanly_or_run():
    import torch
    ndev = torch.cuda.device_count()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", BENCH, "--gpus", "2", "--nx", "1024", "--ny", "512", "--steps", "20", "--warmup", "4", "--cpu-steps", "0"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if ndev >= 2:
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads(lines[-1])
        assert d["n_gpus"] == 2 and len(d["device_ms"]) == 2 and d["value"] > 0
        # first-contact kit: what every rank saw (the same keys the one-GPU `--local-slabs` line carries, tests/test_gpu_bench_contract.py)
        assert d["comm_ranks_seen"] == [2] and [r["rank"] for r in d["ranks"]] == [0, 1]
        for r in d["ranks"]:
            assert r["comm_ranks"] == 2 and r["exchanges"] >= 1 and r["exchange_ms_each"] > 0 and r["agree_checks"] >= 1
    else:
        assert p.returncode != 0
        assert not lines                                            # no bench line from a failed job
        assert "[bench.py] rank 1 (device 1) FAILED" in p.stderr and "HIP device(s) are visible" in p.stderr
        assert "NCCL_DEBUG=WARN" in p.stderr
