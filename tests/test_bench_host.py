"""bench.py's host logic that needs no GPU: the BASELINE configurations behind --config, the workload label, the selection of the pass's kernels
out of a counter table, the measured vector-ALU fraction and the bound that follows from it."""
import importlib.util
import os
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(bench, *argv):
    old = sys.argv
    sys.argv = ["bench.py", *argv]
    try:
        return bench.parse_args()
    finally:
        sys.argv = old


def test_every_baseline_configuration_runs_on_its_own_parameters(bench):
    a = _args(bench)                                      # default: the configuration the metric is quoted on
    assert (a.config, a.nx, a.ny, a.dtype, a.shape, a.aoa, a.tau, a.steps, a.warmup, a.custom) == (2, 4096, 4096, "float32", "naca6409", 10.0, 0.58, 400, 300, [])
    assert "BASELINE configs[2]" in bench.workload_name(a, a.nx, a.ny, "NACA6409")
    a = _args(bench, "--config", "0")
    assert (a.nx, a.ny, a.shape, a.aoa, a.steps) == (256, 128, "naca0012", 0.0, 500) and "BASELINE configs[0]" in bench.workload_name(a, 256, 128, "NACA0012")
    a = _args(bench, "--config", "1")
    assert (a.nx, a.ny, a.shape, a.aoa, a.steps) == (1024, 512, "naca2412", 5.0, 2000)
    a = _args(bench, "--config", "3")
    assert (a.nx, a.ny, a.shape, a.aoa, a.dtype) == (16384, 4096, "naca0012", 8.0, "float32")
    a = _args(bench, "--config", "4")
    assert (a.nx, a.ny, a.shape, a.aoa, a.dtype) == (4096, 2048, "naca4412", 12.0, "float64")
    assert a.tau == pytest.approx(0.5004007, abs=1e-7) and a.custom == []          # tau from Re = 1e6 (SURVEY 8d cfg 5)
    # the driver's arguments leave the workload alone; anything else makes it "custom" and says what differs
    a = _args(bench, "--gpus", "1", "--steps", "20", "--warmup", "5")
    assert a.custom == [] and a.steps == 20 and a.warmup == 5
    a = _args(bench, "--nx", "1024", "--ny", "512")
    w = bench.workload_name(a, 1024, 512, "NACA6409")
    assert a.custom == ["nx", "ny"] and "custom workload" in w and "BASELINE configs[2]:" not in w
    a = _args(bench, "--config", "4", "--tau", "0.58")
    assert a.custom == ["tau"]
    a = _args(bench, "--re", "5e5")
    assert a.tau == pytest.approx(0.5 + 3 * 0.06 * (4096 / 1.84) / 5e5) and "tau" in a.custom
    with pytest.raises(SystemExit):
        _args(bench, "--tau", "0.6", "--re", "1e6")
    with pytest.raises(SystemExit):
        _args(bench, "--fuse-sites", "4")                 # the library has refused 4 since round 3


def _table():
    k4 = "wt::k_march3<float,2,4,false,1>"
    return {k4: {"FETCH_SIZE": 800000.0, "WRITE_SIZE": 640000.0, "SQ_ACTIVE_INST_VALU": 1.2e8, "SQ_WAVE_CYCLES": 2.9e8, "SQ_WAIT_ANY": 7.5e7,
                 "SQ_BUSY_CYCLES": 2.2e7, "GRBM_GUI_ACTIVE": 6.4e6},
            "wt::k_halo4<float,2,1>": {"FETCH_SIZE": 34000.0, "WRITE_SIZE": 13000.0, "SQ_ACTIVE_INST_VALU": 4.0e6, "SQ_WAVE_CYCLES": 1.0e7,
                                       "SQ_WAIT_ANY": 3.0e6, "SQ_BUSY_CYCLES": 1.0e6, "GRBM_GUI_ACTIVE": 4.0e5},
            "wt::k_march3<float,2,4,true,1>": {"FETCH_SIZE": 9e9, "WRITE_SIZE": 9e9, "SQ_ACTIVE_INST_VALU": 9e12},       # the emitting pass: not part of a launch
            "wt::k_step<float,false,3>": {"FETCH_SIZE": 315000.0, "WRITE_SIZE": 590000.0}}


def test_traffic_and_valu_fraction_come_from_the_pass_kernels(bench):
    t = bench.select_traffic(_table(), True, 4)
    assert t["hbm_bytes_per_launch"] == (800000.0 + 34000.0) * 1024 * 2 + (640000.0 + 13000.0) * 1024
    assert "WT_TUNE=0" in t["measured"]                   # the children run the modelled cut, and the entry says so
    t1 = bench.select_traffic(_table(), False, 1)
    assert t1["hbm_bytes_per_launch"] == 315000.0 * 1024 * 2 + 590000.0 * 1024
    assert bench.select_traffic({}, True, 4) is None and bench.select_traffic(None, True, 4) is None
    v = bench.select_valu(_table(), True, 4, 1024, 0.35)
    # SQ_ACTIVE_INST_VALU counts quad-cycles; the launch's duration in clocks = GRBM_GUI_ACTIVE / 8 (summed over the XCDs by rocprofv3)
    assert v["valu_busy_frac"] == pytest.approx((1.2e8 + 4.0e6) * 4 / (1024 * (6.4e6 + 4.0e5) / 8))
    assert v["valu_busy_frac_at_2p4_ghz"] == pytest.approx((1.2e8 + 4.0e6) * 4 / (1024 * 0.35e-3 * 2.4e9))
    assert v["valu_active_per_wave"] == pytest.approx((1.2e8 + 4.0e6) / (2.9e8 + 1.0e7)) and 0 < v["waves_waiting_frac"] < 1
    assert bench.select_valu({"wt::k_step<float,false,3>": {"FETCH_SIZE": 1.0}}, False, 1, 1024, 0.2) is None       # no SQ pass: no figure
    # a three-step pass on a four-step plan (a tau without a proved fast division) still sums k_halo4
    use, _ = bench.pass_kernels(True, 3)
    assert use("wt::k_march3<float,2,3,false,0>") and use("wt::k_halo4<float,2,0>") and use("wt::k_halo3<float,2,0>") and not use("wt::k_march3<float,2,3,true,0>")


def test_default_halo_is_clamped_to_the_narrowest_slab():
    """the hosts' default ghost depth (29 in round 4, 61 since round 5) must not exceed what the narrowest slab owns (wt_create_slab refuses that)."""
    sys.path.insert(0, ROOT)
    from airfoil_cfd_tool_amd.distributed import DEFAULT_HALO, default_halo, slab_bounds
    assert DEFAULT_HALO == 61
    assert default_halo(slab_bounds(4096, 8)) == 61
    assert default_halo(slab_bounds(128, 8)) == 16                      # 8 slabs of 16 columns: the default used to fail here
    assert default_halo(slab_bounds(100, 8)) == 12
    assert default_halo(slab_bounds(4096, 3, [0, 10, 2000, 4096])) == 10
    assert default_halo(slab_bounds(128, 8), 32) == 32                  # an explicit value is the caller's (and fails loudly in wt_create_slab)


def test_exchange_cost_model_of_the_rank_report():
    """bench.py's stated exchange model: 9 populations x halo columns x pitch x element size per side and refresh, one xGMI link per
    side at 76.8 GB/s per direction + 12 us fixed; an edge slab exchanges on one side, an inner slab on two (in parallel: the model is per side)."""
    import bench
    m = bench.exchange_model(29, 4096, 4, 2)
    assert m["exchange_bytes_each"] == 9 * 29 * 4096 * 4 == 4276224 and m["exchange_sides"] == 2
    assert abs(m["exchange_model_us"] - (12.0 + 4276224 / 76.8e3)) < 1e-9 and 67.0 < m["exchange_model_us"] < 68.5
    assert bench.exchange_model(29, 1000, 8, 1)["exchange_bytes_each"] == 9 * 29 * 1024 * 8            # the pitch: NY rounded to 256 rows
    assert bench.exchange_model(16, 4096, 4, 0)["exchange_model_us"] == 0.0                               # a one-slab tunnel exchanges nothing


def test_bound_is_mixed_when_the_two_fractions_are_close():
    """valu_busy_frac saturates near 0.9 and is normalised by it; two fractions within 0.1 name no single bound."""
    import bench
    src = open(bench.__file__).read()
    assert '"mixed" if abs(valu_cmp - hbm_cmp) < 0.1' in src and "valu_frac / 0.9" in src
