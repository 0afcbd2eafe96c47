# Top-level build: librmd.so (HIP kernels + C-ABI shim, gfx950), the C++ host harness, the oracle.
#   make lib      -> raymarchdenoisercuda_amd/lib/librmd.so
#   make host     -> build/main  (the reference's `main -t [label]` CLI over the C ABI)
#   make oracle   -> oracle/liboracle.so (test infrastructure only)
#   make experiments -> build/variants/librmd_experiments.so: the same library + the kernels that were measured and lost
#                    (-DRMD_EXPERIMENTS; select with RMD_LIB_PATH=...; their tests are marked `experiments`)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
PKG      := raymarchdenoisercuda_amd
CSRC     := $(PKG)/csrc
LIBDIR   := $(PKG)/lib
LIB      := $(LIBDIR)/librmd.so

HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable \
            -Wno-unused-but-set-variable
KERNELS  := $(CSRC)/runtime.hip $(CSRC)/box_filter.hip $(CSRC)/weighted_filter.hip $(CSRC)/svgf_temporal.hip $(CSRC)/svgf_variance.hip \
            $(CSRC)/svgf_atrous.hip $(CSRC)/svgf_frame.hip $(CSRC)/convert_synth.hip $(CSRC)/strips.hip
OBJS     := $(patsubst $(CSRC)/%.hip,build/%.o,$(KERNELS))

HOSTSRC  := $(wildcard $(PKG)/host/*.cpp)
HOSTFLAGS:= -O2 -std=c++17 -Wall -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__

all: lib oracle host

lib: $(LIB)

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/rmd_api.h Makefile $(wildcard $(CSRC)/*.inc) $(wildcard $(CSRC)/experiments/*.inc)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

oracle:
	$(MAKE) -C oracle

host: build/main

build/main: $(HOSTSRC) $(LIB) $(wildcard include/*.h)
	@mkdir -p build
	$(CXX) $(HOSTFLAGS) -o $@ $(HOSTSRC) -L$(LIBDIR) -lrmd -lz -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

experiments:
	tools/build_variant.sh experiments -DRMD_EXPERIMENTS

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all lib oracle host experiments clean
