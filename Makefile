# Native build of the whole repository (all outputs stay in-tree).
#
#   make            host library, HIP library (gfx950), oracle, native test harness
#   make hip        only path_tracing_amd/lib/libpt_hip.so
#
# hipcc cross-compiles for gfx950 without a GPU present.
ROCM     ?= /opt/rocm
HIPCC    ?= $(ROCM)/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950

PKG      := path_tracing_amd
LIBDIR   := $(PKG)/lib
STRICT   := -ffp-contract=off -fno-fast-math
HOSTFLAGS := -std=c++17 -O2 $(STRICT) -fPIC -pthread -Wall -Iinclude
HIPFLAGS := -std=c++17 -O3 $(STRICT) -fhip-fp32-correctly-rounded-divide-sqrt -fPIC \
            --offload-arch=$(ARCH) -Iinclude -I$(PKG)/csrc -Wall -Wno-unused-function

HIP_SRCS := $(PKG)/csrc/pt_hip.hip
HIP_HDRS := $(wildcard $(PKG)/csrc/*.h) include/pt_hip.h

all: host hip oracle native-tests

host: $(LIBDIR)/libpt_host.so
hip: $(LIBDIR)/libpt_hip.so
oracle:
	$(MAKE) -C oracle -s

$(LIBDIR)/libpt_host.so: $(PKG)/host/scene_build.cpp include/pt_host.h include/pt_hip.h
	mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(PKG)/host/scene_build.cpp

$(LIBDIR)/libpt_hip.so: $(HIP_SRCS) $(HIP_HDRS)
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS)

# Lane-level emulation of the device traversal on the CPU (tests only).
native-tests: tests/native/_build/libtraverse_emul.so
tests/native/_build/libtraverse_emul.so: tests/native/traverse_emul.cpp $(PKG)/csrc/traverse_core.h include/pt_hip.h
	mkdir -p tests/native/_build
	$(CXX) $(HOSTFLAGS) -I$(PKG)/csrc -shared -o $@ tests/native/traverse_emul.cpp

clean:
	rm -rf $(LIBDIR) tests/native/_build
	$(MAKE) -C oracle clean

.PHONY: all host hip oracle native-tests clean
