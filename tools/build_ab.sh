#!/bin/bash
# variants of the library for A/B experiments (tools/ab/, git-ignored): lib_clocks.so = per-unit clocks (tools/unit_clocks.py)
set -e
mkdir -p tools/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -DWT_EXPERIMENT_KNOBS -Wno-unused-function -shared"
/opt/rocm/bin/hipcc $F -DWT_UNIT_CLOCKS -o tools/ab/lib_clocks.so airfoil-cfd-tool_amd/csrc/windtunnel.hip -ldl -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=airfoil-cfd-tool_amd/csrc/libwindtunnel.map
/opt/rocm/bin/hipcc $F -DWT_UNIT_CLOCKS -DWT_CLOCK_REALTIME -o tools/ab/lib_timeline.so airfoil-cfd-tool_amd/csrc/windtunnel.hip -ldl -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=airfoil-cfd-tool_amd/csrc/libwindtunnel.map
/opt/rocm/bin/hipcc $F -o tools/ab/lib_knobs.so airfoil-cfd-tool_amd/csrc/windtunnel.hip -ldl -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=airfoil-cfd-tool_amd/csrc/libwindtunnel.map
