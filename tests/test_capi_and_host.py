"""C-ABI surface and host logic that need no GPU."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_exports():
    with open(os.path.join(ROOT, "include", "windtunnel.h")) as fh:
        txt = fh.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wt_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = _declared_exports()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libwindtunnel.so lacks {n}"
    from airfoil_cfd_tool_amd._capi import EXPORTS
    assert sorted(EXPORTS) == names
    assert b"libwindtunnel" in lib.wt_version()


def test_dynamic_symbol_table_is_the_header(pkg):
    """Built with -fvisibility=hidden and a version script (csrc/libwindtunnel.map): the library's dynamic symbols are exactly the entry
    points include/windtunnel.h declares — no kernel stubs, no libstdc++ instantiations, no C++ internals."""
    import subprocess
    from airfoil_cfd_tool_amd._capi import LIB_PATH
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], check=True, capture_output=True, text=True).stdout
    syms = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert syms == _declared_exports(), sorted(set(syms) ^ set(_declared_exports()))
    assert len(syms) == 32


def test_streamlit_image_keyword_follows_the_installed_version():
    """streamlit_page._full_width_kwarg: `use_container_width` where Streamlit's image() has it, the older keyword otherwise."""
    from airfoil_cfd_tool_amd.streamlit_page import _full_width_kwarg

    def new(img, caption=None, use_column_width=None, use_container_width=False): ...
    def old(img, caption=None, use_column_width=None): ...
    def bare(img): ...
    assert _full_width_kwarg(new) == {"use_container_width": True}
    assert _full_width_kwarg(old) == {"use_column_width": True}
    assert _full_width_kwarg(bare) == {}


def test_argument_errors_without_a_gpu(pkg):
    lib = pkg.load_library()
    h = ctypes.c_void_p()
    assert lib.wt_create(2, 2, 0, 0, ctypes.byref(h)) == -1          # WT_ERR_ARG before any device call
    assert b"3x3" in lib.wt_last_error()
    assert lib.wt_create(64, 64, 7, 0, ctypes.byref(h)) == -1
    assert lib.wt_create_slab(64, 64, 0, 0, 3, 2, 1, ctypes.byref(h)) == -1
    assert lib.wt_create_slab(64, 64, 0, 0, 0, 2, 0, ctypes.byref(h)) == -1
    assert lib.wt_destroy(None) == 0
    assert lib.wt_step(None, 1, 0.58, 0.06) == -1
    assert lib.wt_sync(None) == -1


def test_no_cpu_fallback(pkg):
    """Without a HIP device the product path must fail loudly (no oracle / CPU route)."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    with pytest.raises(pkg.WTError) as ei:
        pkg.Engine(64, 64)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)
    import airfoil_cfd_tool_amd.windtunnel as wtmod
    import airfoil_cfd_tool_amd._capi as capi
    for mod in (wtmod, capi, pkg.geometry):
        with open(mod.__file__) as fh:
            src = fh.read()
        assert not re.search(r"^\s*(import|from)\s+\S*(lbm_numpy|lbm_c\b|oracle)", src, flags=re.M)


def test_stall_label_thresholds(pkg, oracle_np):
    """html:869-884."""
    for frac, want in ((0.0, "Attached"), (0.044, "Attached"), (0.045, "5% sep"), (0.2449, "24% sep"),
                       (0.245, "STALL ≈ 25% sep"), (0.7, "STALL ≈ 70% sep")):
        assert pkg.stall_label(frac) == want
        assert oracle_np.stall_label(frac) == want


def test_reynolds_and_tau(pkg):
    assert round(pkg.reynolds(0.06, 320, 0.58)) == 391
    tau = pkg.tau_from_reynolds(1e6, 0.06, 4096)
    assert tau == pytest.approx(0.5004007, abs=1e-7)                   # SURVEY §8d cfg 5
    assert pkg.reynolds(0.06, 4096, tau) == pytest.approx(1e6, rel=1e-9)
    with pytest.raises(ValueError):
        pkg.tau_from_reynolds(0.0, 0.06, 320)


def test_every_option_the_library_takes_is_documented_in_the_header():
    """wt_set_option's names (csrc/windtunnel.hip) against the option list of include/windtunnel.h: an option a maintainer cannot read about does not
    exist for them (round 5 added window_overlap, fast_div_two_op, selftest_tau, refresh = 2 through this door, no entry point)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "airfoil-cfd-tool_amd", "csrc", "windtunnel.hip")).read()
    hdr = open(os.path.join(root, "include", "windtunnel.h")).read()
    body = src[src.index('extern "C" int wt_set_option'):src.index('extern "C" int wt_get_option')]
    names = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\) == 0', body))
    assert {"window_overlap", "refresh", "fuse_steps", "trim_ghosts", "fast_div_two_op"} <= names
    missing = sorted(n for n in names if f'"{n}"' not in hdr)
    assert missing == [], missing
