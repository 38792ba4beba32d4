"""Developer A/B: bench.py against another build of the library (tools/ab/<WT_AB_LIB>), same process set-up.
    WT_AB_LIB=lib_x.so python3 tools/ab_bench.py --nx 544 --ny 4096 --cpu-steps 0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import airfoil_cfd_tool_amd._capi as capi
lib = os.environ.get("WT_AB_LIB")
if lib:
    capi.LIB_PATH = os.path.join(ROOT, "tools", "ab", lib)
    capi.load_library(capi.LIB_PATH)
import bench
bench.main()
