# Build everything in-tree (the built .so files travel to the GPU box with the snapshot).
#   make            -> front-end lib, HIP engine lib, turbo CLI, oracle
#   make oracle     -> oracle/liboracle.so only (test infrastructure)
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CXX ?= g++
CC ?= gcc
ARCH ?= gfx950

LIBDIR := turbo_amd/lib
BINDIR := turbo_amd/bin
FRONT_SRC := turbo_amd/csrc/front/fzn_parser.cpp turbo_amd/csrc/front/tcn_lower.cpp turbo_amd/csrc/front/simplify.cpp turbo_amd/csrc/front/xcsp3_reader.cpp turbo_amd/csrc/front/front_capi.cpp
FRONT_HDR := turbo_amd/csrc/front/fzn_ast.hpp turbo_amd/csrc/front/tcn.hpp include/turbo_front.h include/turbo_hip.h
HIP_SRC := turbo_amd/csrc/hip/engine.hip
HIP_HDR := $(wildcard turbo_amd/csrc/hip/*.hpp) include/turbo_hip.h
HOST_SRC := $(wildcard turbo_amd/csrc/host/*.cpp)
HOST_HDR := $(wildcard turbo_amd/csrc/host/*.hpp)

CXXFLAGS := -O2 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter
# -amdgpu-atomic-optimizer-strategy=None: the engine's LDS atomics are issued by one lane on purpose; the optimizer's
# wave-aggregation prologue (mbcnt/bcnt) around each of them costs the event loop 3-5 %
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-parameter -Wno-bitwise-instead-of-logical -mllvm -amdgpu-atomic-optimizer-strategy=None $(EXTRA_HIPFLAGS)

all: front hip tuning cli oracle

front: $(LIBDIR)/libturbo_front.so
hip: $(LIBDIR)/libturbo_hip.so
cli: $(BINDIR)/turbo
oracle:
	$(MAKE) -C oracle

$(LIBDIR)/libturbo_front.so: $(FRONT_SRC) $(FRONT_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(FRONT_SRC)

# The engine's ~90 kernel instantiations are dealt to nine translation units (turbo_amd/csrc/hip/kernel_units.hpp): `make -j8` builds a library in the time of its
# slowest unit (~1.5 min) instead of 4-6 minutes of one core.  Objects under build/obj/<variant>/ (scratch, not shipped); the .so files are what travels.
UNITS := 1 2 3 4 5 6 7 8 9
HIP_DIR := turbo_amd/csrc/hip
HIP_DEP := $(HIP_HDR) $(HIP_DIR)/kernel_units.inc
define HIP_VARIANT  # $(1) = variant name, $(2) = extra flags, $(3) = library file
build/obj/$(1)/engine.o: $(HIP_SRC) $$(HIP_DEP)
	@mkdir -p build/obj/$(1)
	$$(HIPCC) $$(HIPFLAGS) $(2) -c -o $$@ $(HIP_SRC)
build/obj/$(1)/unit_%.o: $(HIP_DIR)/unit_%.hip $$(HIP_DEP)
	@mkdir -p build/obj/$(1)
	$$(HIPCC) $$(HIPFLAGS) $(2) -c -o $$@ $$<
$(3): build/obj/$(1)/engine.o $$(foreach n,$$(UNITS),build/obj/$(1)/unit_$$(n).o)
	@mkdir -p $$(dir $$@)
	$$(HIPCC) $$(HIPFLAGS) -shared -o $$@ $$^
endef
$(eval $(call HIP_VARIANT,hip,,$(LIBDIR)/libturbo_hip.so))

# the same engine with the device-side tuning / profiling knobs of tb_config.reserved[0] compiled in (scripts/*_probe.py;
# select it with TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so)
tuning: $(LIBDIR)/libturbo_hip_tuning.so
$(eval $(call HIP_VARIANT,tuning,-DTB_TUNING,$(LIBDIR)/libturbo_hip_tuning.so))

# A/B variants of the engine: `make -j8 variant NAME=foo FLAGS="-DTB_SOMETHING=1"` -> turbo_amd/lib/ab/foo.so (objects under build/obj/foo/; select it with TURBO_HIP_LIB)
ifdef NAME
variant: $(LIBDIR)/ab/$(NAME).so
$(eval $(call HIP_VARIANT,$(NAME),$(FLAGS),$(LIBDIR)/ab/$(NAME).so))
endif

# the same engine with a range check in front of every index the kernels form from host-packed fields (kernels.hpp: TB_BOUNDS): a report
# {site, index, limit, workgroup} instead of a memory fault.  scripts/bounds_soak.py runs the bench workloads and the fuzz families on it.
# (one translation unit: the report lives in `__device__` variables the host shim and every kernel share)
bounds: $(LIBDIR)/libturbo_hip_bounds.so
$(LIBDIR)/libturbo_hip_bounds.so: $(HIP_SRC) $(HIP_DEP)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DTB_BOUNDS -DTB_SINGLE_TU -shared -o $@ $(HIP_SRC)

# one library per phase of the event kernels, each the production kernels with exactly that phase executed twice (compile-time choice:
# scripts/phase_budget.py measures the difference in SQ_INSTS_* to the production library)
PHASES := 1 2 3 4 5 6 7 8 9 10 11 12 13 14
phases: $(foreach n,$(PHASES),$(LIBDIR)/phases/phase_$(n).so)
$(LIBDIR)/phases/phase_%.so: $(HIP_SRC) $(HIP_DEP)
	@mkdir -p $(LIBDIR)/phases
	$(HIPCC) $(HIPFLAGS) -DTB_DOUBLE_PHASE=$* -DTB_SINGLE_TU -shared -o $@ $(HIP_SRC)

$(BINDIR)/turbo: $(HOST_SRC) $(HOST_HDR) $(LIBDIR)/libturbo_front.so $(LIBDIR)/libturbo_hip.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) -fPIE -o $@ $(HOST_SRC) -Iinclude -L$(LIBDIR) -lturbo_front -lturbo_hip -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,$(ROCM)/lib -lpthread

# AddressSanitizer + UBSan on the CPU code (front-end and oracle) over every benchmark input; GPU sanitizers are not available
SAN := -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer
sanitize:
	@mkdir -p build
	$(CC) $(SAN) -std=c11 -c oracle/oracle.c -o build/oracle_san.o
	$(CXX) $(SAN) -std=c++17 tests/tools/san_front.cpp $(FRONT_SRC) -o build/san_front
	$(CXX) $(SAN) -std=c++17 tests/tools/san_oracle.cpp $(FRONT_SRC) build/oracle_san.o -o build/san_oracle
	build/san_front benchmarks/*.fzn benchmarks/test_data/*.fzn benchmarks/test_data/*.xml tests/golden/xcsp3/*.xml benchmarks/unsolved_bugs_data/*.fzn > build/san_front.log
	build/san_oracle benchmarks/*.fzn benchmarks/test_data/*.fzn benchmarks/unsolved_bugs_data/bigdom.fzn > build/san_oracle.log
	@echo "sanitize: no report"

clean:
	rm -rf $(LIBDIR) $(BINDIR)
	$(MAKE) -C oracle clean

.PHONY: all front hip cli oracle sanitize clean tuning phases bounds variant
