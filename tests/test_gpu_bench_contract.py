"""GPU: bench.py prints ONE JSON line with the keys the driver's contract names (task description,
"Measurement"), incl. the `roofline` and `cpu_baseline` objects; checked on a small lattice."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "1024", "--ny", "512", "--steps", "20",
                          "--warmup", "3", "--cpu-steps", "1", *extra], capture_output=True, text=True, timeout=600, check=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract():
    d = _run()
    with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as fh:
        assert d["metric"] == json.load(fh)["metric"]
    for key in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "MLUPS" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    # value and the roofline figure describe the same run: MLUPS * 72 B = GB/s (up to wall-vs-device timing)
    assert 0.5 < (d["value"] * 72 / 1000.0) / r["achieved"] < 1.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "MLUPS" and c["value"] > 0 and "sample" in c
    assert d["value"] > 50 * c["value"]


def test_bench_fused_flag_and_fp64():
    d = _run("--fuse", "2", "--dtype", "float32")           # forced on a small lattice: three steps per pass (the fp32 default)
    assert d["config"]["fuse_steps"] == 1 and d["config"]["fuse_depth"] == 3 and d["roofline"]["steps_per_launch"] == 3
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 3 * 72 * 1024 * 512
    d = _run("--fuse", "2", "--fuse-depth", "2", "--dtype", "float32")
    assert d["config"]["fuse_depth"] == 2 and d["roofline"]["algorithmic_bytes_per_launch"] == 2 * 72 * 1024 * 512
    d = _run("--fuse", "0")
    assert d["config"]["fuse_steps"] == 0 and d["roofline"]["algorithmic_bytes_per_launch"] == 72 * 1024 * 512
    d = _run("--dtype", "float64")
    assert d["dtype"] == "f64" and d["roofline"]["algorithmic_bytes_per_launch"] == 144 * 1024 * 512
