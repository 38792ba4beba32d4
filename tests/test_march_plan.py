"""The unit planner of the marching kernels (build_march_plan, csrc/step_march.hpp) on the host: every marched column in exactly
one unit, length caps, the minimum length of a window's last unit for four-step passes, outlet flags, slab ranges."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not present")
def test_march_plan_invariants(tmp_path):
    exe = tmp_path / "plan_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-o", str(exe),
                    os.path.join(ROOT, "tests", "_march_plan_check.hip")], check=True, capture_output=True, timeout=900)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "plan check passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
