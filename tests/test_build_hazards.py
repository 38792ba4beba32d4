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


def test_marching_loop_instruction_budget(isa_files):
    """The four-step fp32 kernel is bound by the vector instructions two waves per SIMD can issue, so the size of its loop is pinned here,
    on the CPU box, where a regression costs nothing to catch: instructions EXECUTED by one trip round the chain loop on the
    path of a plain interior column (tools/isa_loops.py follows the branches a column without body, clamp or far-field rows takes), per wave =
    4 levels x 128 rows.  Round 4: 1096 executed / 867 vector (536 packed, 104 plain moves); round 5: 1030 / 805 (496 packed, 82 moves) with the
    two-operation division by tau, 1066 / 841 with the three-operation one; fp64: 1211 / 1021 per 4 x 64 rows (round 4: 9 IEEE divisions per site more)."""
    import isa_loops as L
    f = [p for p in isa_files if "windtunnel" in os.path.basename(p)][0]
    budget = {   # kernel symbol -> (executed, vector, packed, plain moves) upper bounds: the round-5 counts + 1 % of slack for compiler noise
        "_ZN2wt8k_march3IfLi2ELi4ELb0ELi17EEEvNS_11MarchParamsIT_EE": (1042, 814, 496, 90),
        "_ZN2wt8k_march3IfLi2ELi4ELb0ELi1EEEvNS_11MarchParamsIT_EE": (1078, 850, 532, 90),
        "_ZN2wt8k_march3IdLi1ELi4ELb0ELi1EEEvNS_11MarchParamsIT_EE": (1225, 1032, 0, 56),
        # overlapping windows (FD bit 5; slabs and small whole lattices): no halo-line load, six data-parallel moves per stage instead of twelve, no seam rows — 985 / 785
        "_ZN2wt8k_march3IfLi2ELi4ELb0ELi49EEEvNS_11MarchParamsIT_EE": (996, 794, 496, 90),
    }
    for sym, (ex_max, valu_max, pk_max, mov_max) in budget.items():
        name, lines = L.kernel_lines(f, sym)
        assert name == sym, sym
        best = None
        for tgt, a, b, c, nops in L.loops(lines):
            if not (800 < sum(c.values()) < 4000):
                continue
            try:
                cc, _ = L.follow(lines, tgt)
            except (KeyError, IndexError):
                continue
            tot = sum(cc.values())
            if 700 < tot < 1300 and cc.get("vmem", 0) == (18 if "ELi49E" in sym else 21) and (best is None or tot < sum(best.values())):
                best = cc                      # the chain units' loop: 9 loads + 1 halo line + 9 stores + 2 seam stores (overlapping windows: 9 + 9)
        assert best is not None, f"no marching loop found in {sym}"
        valu = sum(v for k, v in best.items() if k.startswith("v_"))
        got = (sum(best.values()), valu, best.get("v_pk", 0), best.get("v_mov", 0))
        assert got[0] <= ex_max and got[1] <= valu_max and got[2] <= pk_max and got[3] <= mov_max, (sym, got, (ex_max, valu_max, pk_max, mov_max))
