# Builds libwindtunnel.so (HIP, gfx950) in-tree and the C oracle (test infrastructure).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG       = airfoil-cfd-tool_amd
CSRC      = $(PKG)/csrc
LIB       = $(PKG)/lib/libwindtunnel.so
# -ffp-contract=off: one rounding per operation, as the oracle (and the parity tests) assume.
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function
# EXPERIMENT=1: the planner / launch-order environment knobs of tools/ (include/windtunnel.h "Environment"); never for a production build
ifeq ($(EXPERIMENT),1)
HIPFLAGS += -DWT_EXPERIMENT_KNOBS
endif
# No -lrccl: the RCCL entry points are bound at the first wt_comm_* call to the ONE RCCL the process has mapped (csrc/rccl_bind.hpp); the run path is
# for processes without PyTorch, where that call has to dlopen librccl.so.1 itself.
LDFLAGS  ?= -ldl -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=$(CSRC)/libwindtunnel.map

all: lib oracle

lib: $(LIB)

$(LIB): $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.hpp) $(CSRC)/libwindtunnel.map include/windtunnel.h
	mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/windtunnel.hip $(LDFLAGS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(PKG)/lib oracle/_build
.PHONY: all lib oracle clean
