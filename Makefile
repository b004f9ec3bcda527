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
# -fno-slp-vectorize: packed-f32 pairs cost the trace kernels more register shuffles than they
# save arithmetic (measured: 55 instead of 68 VGPRs and 5-15 % faster without them).
HIPFLAGS := -std=c++17 -O3 $(STRICT) -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC \
            --offload-arch=$(ARCH) -Iinclude -I$(PKG)/csrc -Wall -Wno-unused-function

HIP_SRCS := $(PKG)/csrc/pt_hip.hip
HIP_HDRS := $(wildcard $(PKG)/csrc/*.h) include/pt_hip.h

all: host hip render oracle native-tests

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

# The reference's C++ scene/render API on top of the C ABI, and its command-line front end.
render: $(LIBDIR)/libpt_render.so $(PKG)/bin/cli_render $(PKG)/bin/animate
$(LIBDIR)/libpt_render.so: $(PKG)/host/render_api.cpp include/pt_render.h include/pt_hip.h include/pt_host.h $(LIBDIR)/libpt_host.so $(LIBDIR)/libpt_hip.so
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(PKG)/host/render_api.cpp -L$(LIBDIR) -lpt_host -lpt_hip -lz -Wl,-rpath,'$$ORIGIN'
$(PKG)/bin/cli_render: $(PKG)/host/cli_render.cpp include/pt_render.h $(LIBDIR)/libpt_render.so
	mkdir -p $(PKG)/bin
	$(CXX) $(HOSTFLAGS) -o $@ $(PKG)/host/cli_render.cpp -L$(LIBDIR) -lpt_render -lpt_host -lpt_hip -lz -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath-link,$(ROCM)/lib

$(PKG)/bin/animate: $(PKG)/host/animate.cpp include/pt_render.h $(LIBDIR)/libpt_render.so
	mkdir -p $(PKG)/bin
	$(CXX) $(HOSTFLAGS) -o $@ $(PKG)/host/animate.cpp -L$(LIBDIR) -lpt_render -lpt_host -lpt_hip -lz -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath-link,$(ROCM)/lib

# Lane-level emulation of the device traversal on the CPU (tests only).
native-tests: tests/native/_build/libtraverse_emul.so tests/native/_build/libpt_hip_smallstack.so tests/native/_build/api_check
# Where the reference's tree is at hand: its own example front ends (main.cpp, animate.cpp), compiled
# UNMODIFIED from where they lie -- read from standard input, so that their `#include "integrator.h"` /
# `"visualizer.h"` find the two forwarding headers written here (one line each: #include "pt_render.h",
# the include swap a maintainer makes) -- and linked against the libraries above.  The drop-in check of
# the C++ surface; outputs stay in the (git-ignored) build directory.
REF ?= /root/reference
ifneq ($(wildcard $(REF)/main.cpp),)
native-tests: tests/native/_build/ref_main tests/native/_build/ref_animate
tests/native/_build/ref_%: $(REF)/%.cpp include/pt_render.h $(LIBDIR)/libpt_render.so
	mkdir -p tests/native/_build/fwd
	echo '#include "pt_render.h"' > tests/native/_build/fwd/integrator.h
	echo '#include "pt_render.h"' > tests/native/_build/fwd/visualizer.h
	$(CXX) $(HOSTFLAGS) -iquote tests/native/_build/fwd -x c++ - -o $@ -L$(LIBDIR) -lpt_render -lpt_host -lpt_hip -lz -Wl,-rpath,'$$ORIGIN/../../../$(LIBDIR)' -Wl,-rpath-link,$(ROCM)/lib < $<
endif
# Exercises the C++ API directly (tests/test_gpu_cpp_api.py).
tests/native/_build/api_check: tests/native/api_check.cpp include/pt_render.h $(LIBDIR)/libpt_render.so
	mkdir -p tests/native/_build
	$(CXX) $(HOSTFLAGS) -o $@ tests/native/api_check.cpp -L$(LIBDIR) -lpt_render -lpt_host -lpt_hip -lz -Wl,-rpath,'$$ORIGIN/../../../$(LIBDIR)' -Wl,-rpath-link,$(ROCM)/lib
# The same HIP library with only 4 stack entries per lane in LDS: nearly every ray then uses the
# spill path to HBM, which the regular build needs for < 1 % of rays (tests/test_gpu_spill.py).
tests/native/_build/libpt_hip_smallstack.so: $(HIP_SRCS) $(HIP_HDRS)
	mkdir -p tests/native/_build
	$(HIPCC) $(HIPFLAGS) -DSTACK_LDS=4 -shared -o $@ $(HIP_SRCS)
tests/native/_build/libtraverse_emul.so: tests/native/traverse_emul.cpp $(PKG)/csrc/traverse_core.h $(PKG)/csrc/qnode_build.h include/pt_hip.h
	mkdir -p tests/native/_build
	$(CXX) $(HOSTFLAGS) -I$(PKG)/csrc -shared -o $@ tests/native/traverse_emul.cpp

clean:
	rm -rf $(LIBDIR) $(PKG)/bin tests/native/_build
	$(MAKE) -C oracle clean

.PHONY: all host hip render oracle native-tests clean
