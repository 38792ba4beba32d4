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
                          "--warmup", "3", "--cpu-steps", "1", *(() if "--pmc-traffic" in extra else ("--pmc-traffic", "0")), *extra],
                         capture_output=True, text=True, timeout=600, check=True)
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
    # (k_step is HBM-bound, the marching kernels are bound by their vector instructions; this lattice sits at the edge of the automatic choice)
    assert r["bound"] == ("valu" if d["config"]["fuse_steps"] else "hbm") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # (--pmc-traffic 0 and) no rocprofv3 counter entry exists for this lattice: `frac` is null, never the effective figure
    assert r["frac"] is None and r["achieved"] is None and r["traffic"] is None
    assert 0 < r["compulsory_frac"] <= 1.0 and abs(r["compulsory_frac"] - r["compulsory_gbps"] / r["peak"]) < 1e-12
    # value and the throughput figure describe the same run: MLUPS * 72 B = GB/s (up to wall-vs-device timing)
    assert 0.5 < (d["value"] * 72 / 1000.0) / r["effective_gbps"] < 1.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "MLUPS" and c["value"] > 0 and "sample" in c
    assert d["value"] > 50 * c["value"]


def test_bench_fused_flag_and_fp64():
    d = _run("--fuse", "2", "--dtype", "float32")           # forced on a small lattice: three steps per pass (the fp32 default)
    assert d["config"]["fuse_steps"] == 1 and d["config"]["fuse_depth"] == 3 and d["roofline"]["steps_per_launch"] == 3
    assert d["roofline"]["bound"] == "valu" and d["roofline"]["frac"] is None and d["roofline"]["compulsory_frac"] <= 1.0
    assert d["config"]["single_steps"] == 0                 # warm-up 3 = one pass, 20 timed steps = 3 x 6 + 2: no single step
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 3 * 72 * 1024 * 512
    d = _run("--fuse", "2", "--fuse-depth", "2", "--dtype", "float32")
    assert d["config"]["fuse_depth"] == 2 and d["roofline"]["algorithmic_bytes_per_launch"] == 2 * 72 * 1024 * 512
    d = _run("--fuse", "0")
    assert d["config"]["fuse_steps"] == 0 and d["roofline"]["algorithmic_bytes_per_launch"] == 72 * 1024 * 512
    d = _run("--dtype", "float64")
    assert d["dtype"] == "f64" and d["roofline"]["algorithmic_bytes_per_launch"] == 144 * 1024 * 512


def test_bench_measures_its_traffic_in_the_same_session():
    """Default behaviour: two short child runs under rocprofv3 --pmc measure the workload's HBM bytes per launch on THIS box just before the timed
    run, for any lattice — here one that has no entry in profiles/pmc_traffic.json."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    d = _run("--pmc-traffic", "1", "--fuse", "2")
    r = d["roofline"]
    assert r["frac"] is not None and 0.05 < r["frac"] <= 1.0, r
    assert "this session" in r["traffic_source"]["measured"]
    # a four-step pass over 1024 x 512 fp32 moves at least one lattice read + one lattice write, and far less than four single steps
    assert 72 * 1024 * 512 <= r["traffic"] <= 2.5 * 72 * 1024 * 512, r["traffic"]


def test_bench_default_frac_is_counter_based_and_below_one():
    """The default workload has a counter entry (profiles/pmc_traffic.json): frac = measured HBM bytes / this run's time <= 1."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-steps", "0", "--pmc-traffic", "0"],
                         capture_output=True, text=True, timeout=600, check=True)
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    r = d["roofline"]
    assert r["frac"] is not None and 0.2 < r["frac"] <= 1.0 and r["traffic"] > 1.2e9
    assert r["single_step"]["frac"] is not None and r["single_step"]["frac"] <= 1.0
    assert d["config"]["single_steps"] == 0 and d["config"]["fuse_depth"] >= 3        # --warmup 5 leaves the seam buffer valid


def test_bench_local_slabs_and_dat(tmp_path):
    dat = tmp_path / "sym.dat"
    import numpy as np
    xs = 0.5 * (1 - np.cos(np.linspace(0, np.pi, 40)))
    yt = 0.6 * (0.2969 * np.sqrt(xs) - 0.1260 * xs - 0.3516 * xs ** 2 + 0.2843 * xs ** 3 - 0.1036 * xs ** 4)
    pts = [(x, y) for x, y in zip(xs[::-1], yt[::-1])] + [(x, -y) for x, y in zip(xs[1:], yt[1:])]
    dat.write_text("test foil\n" + "\n".join(f"{x:.6f} {y:.6f}" for x, y in pts) + "\n")
    d = _run("--dat", str(dat))
    assert "sym.dat" in d["config"]["workload"] and d["config"]["solid_sites"] > 1000
    d = _run("--local-slabs", "4", "--halo", "8", "--fuse", "2")
    ls = d["local_slabs"]
    assert len(ls["device_ms_per_step"]) == 4 and ls["one_slab_alone_ms_per_step"] > 0 and d["config"]["slabs"] == 4
