"""The generated gfx950 ISA must not contain the buffer-store data hazard hipcc leaves unpadded
(store_data_fence() in csrc/step_march.hpp; tools/check_store_hazard.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_store_hazard as chk          # noqa: E402


def test_checker_flags_an_unpadded_overwrite(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n\tbuffer_store_dwordx4 v[4:7], v0, s[40:43], s0 offen nt\n\tv_mov_b32_e32 v5, v9\n\ts_endpgm\n")
    ok = tmp_path / "ok.s"
    ok.write_text("k:\n\tbuffer_store_dwordx4 v[4:7], v0, s[40:43], s0 offen nt\n\ts_nop 1\n\tv_mov_b32_e32 v5, v9\n"
                  "\tbuffer_store_dwordx4 v[4:7], v0, s[40:43], 0 offen\n\tv_mov_b32_e32 v5, v9\n\ts_endpgm\n")
    assert len(chk.check(str(bad))[1]) == 1
    n, b = chk.check(str(ok))
    assert n == 1 and b == []              # the immediate-soffset store is hipcc's to pad, not counted


@pytest.fixture(scope="module")
def isa_files():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not present")
    return chk.build()


def test_no_step_kernel_uses_scratch(isa_files):
    """Every stepping kernel of the library (k_step and the marching kernels) keeps its state in registers: a spilling instantiation is
    not a product-quality path (round 2 shipped six; they are gone, and so are the fp32 four-step kernels with the IEEE division by tau).
    The once-per-pass halo kernels (38-77 VGPRs, no register pressure) may keep up to 48 bytes there: hipcc routes two of the nine results
    of site_step1's four-way branch through the stack."""
    seen, spilling = 0, []
    for f in isa_files:
        for name, r in chk.resources(f).items():
            if not any(k in name for k in ("k_step", "k_march", "k_halo")):
                continue
            seen += 1
            limit = 48 if "k_halo" in name else 0
            if r.get("private_seg_size", 0) > limit:
                spilling.append((name, r))
    assert seen >= 30, seen
    assert spilling == [], spilling


def test_library_isa_has_no_store_data_hazard(isa_files):
    files = isa_files
    assert files
    total, bad = 0, []
    for f in files:
        n, b = chk.check(f)
        total += n
        bad += b
    assert total >= 50                     # the marching kernels' store groups were seen
    assert bad == [], bad[:3]
