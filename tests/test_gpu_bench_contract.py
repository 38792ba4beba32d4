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
    # everything the device did before the declared warm-up is in the line (VERDICT r3 item 3b / ADVICE r3)
    assert d["preheat_steps"] >= 0 and d["preheat_ms"] >= 0 and isinstance(d["preheat_what"], str)
    if d["config"]["fuse_steps"]:
        assert d["preheat_steps"] >= 44 and d["preheat_ms"] > 0
    assert "custom workload" in d["config"]["workload"] and "BASELINE configs[2]:" not in d["config"]["workload"]
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # `bound` follows from fractions MEASURED in the session; with --pmc-traffic 0 nothing is measured: the SURVEY's roof, no vector-ALU figure
    assert r["bound"] == "hbm" and r["valu_busy_frac"] is None and r["valu"] is None and r["unit"] == "GB/s" and r["peak"] == 8000.0
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
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] is None and d["roofline"]["compulsory_frac"] <= 1.0
    assert d["config"]["pass_depth"] == 3
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
    # the SQ group of the third child run: a measured vector-ALU fraction, and `bound` = the larger of the two measured fractions
    v = r["valu"]
    assert v is not None and 0.02 < r["valu_busy_frac"] <= 2.0 and r["valu_busy_frac"] == v["valu_busy_frac"], r
    assert 0 < v["valu_active_per_wave"] <= 1.0 and 0 <= v["waves_waiting_frac"] <= 1.0 and v["SQ_ACTIVE_INST_VALU"] > 0
    assert r["bound"] == ("valu" if r["valu_busy_frac"] > r["frac"] else "hbm")
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
    # the per-rank report an N-GPU line carries (`ranks`), here from the in-process transport
    assert [r["rank"] for r in d["ranks"]] == [0, 1, 2, 3]
    for r in d["ranks"]:
        for key in ("comm_ranks", "device_ms", "exchanges", "exchange_ms_each", "interior_ms_each", "exchange_exposed_ms_each", "exchange_hidden_frac",
                    "fuse_depth", "pass_depth", "passes", "single_steps", "agree_checks", "chain_downgrades", "x0", "width"):
            assert key in r, key
        assert r["exchanges"] >= 1 and r["exchange_ms_each"] > 0 and r["chain_downgrades"] == 0 and r["agree_checks"] >= 1


def test_bench_config_flag_runs_each_baseline_configuration_on_its_own_parameters():
    """--config N: BASELINE.json's configurations on their OWN shape / AoA / tau / dtype / lattice; cfg 0 and cfg 4 here."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "0", "--cpu-steps", "0", "--pmc-traffic", "0", "--steps", "40", "--warmup", "8"],
                         capture_output=True, text=True, timeout=600, check=True)
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert "BASELINE configs[0]" in d["config"]["workload"] and "NACA0012 256x128 float32" in d["config"]["workload"] and d["config"]["solid_sites"] == 1566
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--cpu-steps", "0", "--pmc-traffic", "0", "--steps", "24", "--warmup", "8"],
                         capture_output=True, text=True, timeout=600, check=True)
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert "BASELINE configs[4]" in d["config"]["workload"] and "tau=0.5004007" in d["config"]["workload"] and d["dtype"] == "f64"
    assert d["config"]["solid_sites"] == 405515 and d["config"]["nx"] == 4096 and d["config"]["ny"] == 2048
