# Build without Python: libgfasort_hip.so (hipcc, gfx950), the C++ host mirror / CLI, the oracle.
# Same flags and outputs as gfasort_amd/build.py (which the tests, bench.py and __graft_entry__.py use).
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CSRC    := gfasort_amd/csrc
HOST    := $(CSRC)/host
LIBDIR  := gfasort_amd/lib
OBJDIR  := $(LIBDIR)/obj
BINDIR  := gfasort_amd/bin
LIB     := $(LIBDIR)/libgfasort_hip.so

# -ffp-contract=off : the reference (Rust) never fuses a*b+c; device, host tables and oracle must match it
# -munsafe-fp-atomics: native global_atomic_add_f64 on hipMalloc'ed (coarse-grained) memory
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function
CXXFLAGS := -O2 -std=c++17 -Wall -ffp-contract=off

KERNELS := sgd_kernels_1d sgd_kernels_nd sgd_kernels_nd_team index_kernels capi multi
OBJS    := $(KERNELS:%=$(OBJDIR)/%.o)
HDRS    := $(CSRC)/sgd_device.h $(CSRC)/sgd_kernel_common.h include/gfasort_hip.h
HOSTSRC := $(HOST)/graph.cpp $(HOST)/sgd.cpp
HOSTHDR := $(HOST)/graph.hpp $(HOST)/sgd.hpp

all: $(LIB) $(BINDIR)/gfasort_hip $(BINDIR)/host_selftest $(BINDIR)/multi_rank_selftest oracle

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

$(BINDIR)/gfasort_hip: $(HOST)/main.cpp $(HOSTSRC) $(HOSTHDR) $(LIB)
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp $(HOSTSRC) -L$(LIBDIR) -lgfasort_hip -pthread '-Wl,-rpath,$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

$(BINDIR)/host_selftest: $(HOST)/selftest.cpp $(HOSTSRC) $(HOSTHDR) $(LIB)
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/selftest.cpp $(HOSTSRC) -L$(LIBDIR) -lgfasort_hip -pthread '-Wl,-rpath,$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

$(BINDIR)/multi_rank_selftest: $(HOST)/multi_selftest.cpp $(LIB)
	@mkdir -p $(BINDIR)
	$(HIPCC) -x hip --offload-arch=gfx950 -O2 -std=c++17 -o $@ $(HOST)/multi_selftest.cpp -L$(LIBDIR) -lgfasort_hip -pthread '-Wl,-rpath,$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

selftest: $(BINDIR)/host_selftest
	$(BINDIR)/host_selftest tests/data

clean:
	rm -rf $(OBJDIR) $(LIB) $(BINDIR)

.PHONY: all oracle selftest clean
