"""Runs ON THE GPU BOX against the EXPERIMENT build (tools/build_ab.sh -> tools/ab/lib_clocks.so, -DWT_EXPERIMENT_KNOBS): what happens when an
ill-formed chain plan reaches the device all the same?  WT_DEBUG_CORRUPT_PLAN=1 strips the chain flags of one unit of the first chain block
behind the library's plan guard; its partner then waits for a hand-over that never comes.  Expected: the bounded poll of chain_receive
(step_chain.hpp) ends the wait after CHAIN_POLL_LIMIT polls, the kernel terminates, and wt_sync reports WT_ERR_STATE — no hang."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WT_DEBUG_CORRUPT_PLAN"] = "1"
os.environ["WT_TUNE"] = "0"
import numpy as np
import airfoil_cfd_tool_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", "lib_clocks.so")
capi.load_library(capi.LIB_PATH)
import airfoil_cfd_tool_amd as pkg
nx, ny = 2048, 512
with pkg.Engine(nx, ny) as e:
    e.set_option("fuse_depth", 4); e.set_option("fuse_steps", 2)
    e.set_mask(np.zeros((ny, nx), np.uint8)); e.init_equilibrium(0.06)
    print(f"chain units in the (corrupted) plan: {int(e.get_option('chain_units'))} of {int(e.get_option('fuse_units'))}", flush=True)
    t0 = time.perf_counter()
    e.step(4, 0.58, 0.06)
    try:
        e.sync()
        print(f"sync returned OK after {time.perf_counter() - t0:.2f} s: the corrupted unit did not wait (unexpected)")
    except pkg.WTError as err:
        print(f"kernel terminated after {time.perf_counter() - t0:.2f} s; wt_sync: {err}")
    e.init_equilibrium(0.06)          # clears the flag; the handle is usable again (with its plan still corrupt: do not step it)
    e.sync()
    print("after wt_init_equilibrium: wt_sync OK")
