"""bench.py's N > 1 entry: it must fail CLEANLY and diagnosably — nothing here needs more than the
machine it runs on.

* without a launcher `--gpus 2` exits with status 2 and says how to launch (CPU test: it exits before importing torch);
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


def test_gpus_without_launcher_exits_with_instructions():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 2
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node" in p.stderr
    assert p.stdout.strip() == ""


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
