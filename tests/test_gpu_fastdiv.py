"""The fast divisions by the relaxation time (csrc/d2q9.hpp) against the IEEE quotient, counted on the device.

fp32: the library's own proof — all 2^23 significands, both signs — for the three- and the two-operation form; a tau the two-operation
form is known to fail on (tools/fastdiv_check.c) must be reported as failing, and the stepping kernels must then keep the three-operation
form (same bits).  fp64: the four-operation form (a theorem for every tau) on 2^28 pseudo-random and boundary-hugging numerators per tau.
Every step of STEP_FS divides by tau nine times per site (html:352-356)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAUS = [0.58, 0.5 + 3.0 * 0.06 * (4096 / 1.84) / 1e6, 0.51, 0.6, 0.8168003, 1.0, 1.7]


@pytest.mark.parametrize("tau", TAUS)
def test_fp64_four_operation_division_equals_ieee(pkg, tau):
    with pkg.WindTunnel(shape="naca0012", nx=256, ny=128, dtype="float64") as wt:
        wt.engine.set_option("selftest_tau", tau)
        assert wt.engine.get_option("selftest_fastdiv64") == 0.0


@pytest.mark.parametrize("tau", TAUS)
def test_fp32_three_operation_division_is_proved(pkg, tau):
    with pkg.WindTunnel(shape="naca0012", nx=256, ny=128) as wt:
        wt.engine.set_option("selftest_tau", tau)
        assert wt.engine.get_option("selftest_fastdiv32_3") == 0.0


def test_fp32_two_operation_division_proved_or_refused(pkg):
    """0.58 (the reference's tau, html:78) passes; 0.816800296 (0x3f5119d3) is one of the ~1 % of tau whose two-operation form misses ONE significand."""
    with pkg.WindTunnel(shape="naca0012", nx=256, ny=128) as wt:
        wt.engine.set_option("selftest_tau", 0.58)
        assert wt.engine.get_option("selftest_fastdiv32_2") == 0.0
        bad_tau = float(np.array([0x3f5119d3], dtype=np.uint32).view(np.float32)[0])
        wt.engine.set_option("selftest_tau", bad_tau)
        assert wt.engine.get_option("selftest_fastdiv32_2") > 0.0
        assert wt.engine.get_option("selftest_fastdiv32_3") == 0.0


@pytest.mark.parametrize("tau_bits,two_op", [(0x3f147ae1, 1.0), (0x3f5119d3, 0.0)], ids=["tau0.58", "tau0.8168"])
def test_four_step_kernel_picks_the_proved_form_and_keeps_the_bits(pkg, tau_bits, two_op):
    """The four-step fp32 kernel with the two-operation division (where proved), with the three-operation one and the one-step kernel: the same populations."""
    tau = float(np.array([tau_bits], dtype=np.uint32).view(np.float32)[0])
    out = []
    for fuse, two in ((0, 1), (2, 1), (2, 0)):
        with pkg.WindTunnel(shape="naca2412", nx=512, ny=256, aoa_deg=5.0, tau=tau) as wt:
            wt.engine.set_option("fuse_depth", 4)
            wt.engine.set_option("fuse_chunk", 8)
            wt.engine.set_option("fast_div_two_op", two)
            wt.engine.set_option("fuse_steps", fuse)
            wt.sim_step(48)
            out.append(wt.read_f())
            if fuse:
                assert wt.engine.get_option("fast_div_active") == 1.0
                assert wt.engine.get_option("fast_div_two_op_active") == (two_op if two else 0.0)
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
