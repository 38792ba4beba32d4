"""Which RCCL does libwindtunnel.so talk to?    The library has no link-time dependency on RCCL any more; the first wt_comm_*
call binds the ONE copy the process has mapped (csrc/rccl_bind.hpp) and wt_version() names it; two mapped copies are an error, not a choice.
Each case runs in a process of its own (what is mapped cannot be undone) and leaves through os._exit: no HIP call is made, no GPU is needed."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "airfoil-cfd-tool_amd", "lib", "libwindtunnel.so")
PRELUDE = f"""
import ctypes, os, re, sys
def copies():
    return sorted({{os.path.realpath(l.split()[-1]) for l in open('/proc/self/maps') if re.search(r'/librccl\\.so[.0-9]*$', l.strip())}})
def load():
    lib = ctypes.CDLL({LIB!r}, mode=ctypes.RTLD_GLOBAL)
    lib.wt_version.restype = ctypes.c_char_p
    lib.wt_last_error.restype = ctypes.c_char_p
    return lib
def done(*a):
    print(*a, flush=True)
    os._exit(0)
"""


def run(body):
    if not os.path.exists(LIB):
        pytest.skip("libwindtunnel.so not built")
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    p = subprocess.run([sys.executable, "-c", PRELUDE + body], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout, p.stderr[-2000:])
    return p.stdout


def test_binding_loaded_before_torch_sees_one_rccl():
    out = run("""
lib = load()
assert copies() == [], copies()                       # loading the library maps no RCCL at all
assert b'not bound yet' in lib.wt_version()
import torch
c = copies()
assert len(c) == 1 and '/torch/' in c[0], c           # the only RCCL in the process is the one PyTorch brought
buf = ctypes.create_string_buffer(256)
rc = lib.wt_comm_unique_id(buf)
assert rc == 0, lib.wt_last_error()
v = lib.wt_version().decode()
assert copies() == c and c[0] in v and 'RCCL 2.' in v, (v, copies())
done('OK', v)
""")
    assert out.startswith("OK")


def test_without_torch_the_run_path_copy_is_bound_and_named():
    out = run("""
lib = load()
buf = ctypes.create_string_buffer(256)
rc = lib.wt_comm_unique_id(buf)                        # (ROCm's own RCCL wants a device for this: on a machine without one the CALL fails inside RCCL, -3 —
assert rc in (0, -3), (rc, lib.wt_last_error())        #  the binding, which is what is tested here, has happened either way)
c = copies()
v = lib.wt_version().decode()
assert len(c) == 1 and c[0] in v and 'torch' not in sys.modules, (c, v)
done('OK', v)
""")
    assert out.startswith("OK") and "/opt/rocm" in out


def test_two_mapped_rccl_copies_are_refused_with_both_paths():
    if not os.path.exists("/opt/rocm/lib/librccl.so.1"):
        pytest.skip("no second RCCL on this machine")
    out = run("""
import torch
other = ctypes.CDLL('/opt/rocm/lib/librccl.so.1')      # a second copy, as a process that loaded an -lrccl library before torch used to hold
c = copies()
assert len(c) == 2, c
lib = load()
buf = ctypes.create_string_buffer(256)
rc = lib.wt_comm_unique_id(buf)
msg = lib.wt_last_error().decode()
assert rc == -3 and 'two RCCL libraries' in msg and all(p in msg for p in c), (rc, msg)
assert 'RCCL unavailable' in lib.wt_version().decode()
rc2 = lib.wt_comm_selftest(0, 64)                      # every wt_comm_* entry point says the same, none reaches RCCL
assert rc2 == -3 and 'two RCCL libraries' in lib.wt_last_error().decode()
done('OK')
""")
    assert out.startswith("OK")
