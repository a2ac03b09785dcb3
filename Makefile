# Builds the C-ABI HIP library for gfx950 (cross-compiles without a GPU).
#   make            -> scene-net_amd/lib/libscenenet_hip.so
#   make asm        -> build/*.s + resource usage (VGPR/LDS/occupancy) for inspection
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG      := scene-net_amd
SRC      := $(PKG)/csrc
OUT      := $(PKG)/lib
OBJDIR   := build/obj
# EXTRA: e.g. `make -B EXTRA=-DSN_CONV_TIMING` builds the per-workgroup phase clocks tools/conv_timing.py reads
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(SRC) -Wall -Wno-unused-function $(EXTRA)
# voxel.hip reproduces numpy's fp64 rounding sequence: never contract a*b+c
FLAGS_voxel := -ffp-contract=off

SOURCES := cabi bank voxel conv conv_i8 conv_i8s conv_lin backward corr loss
OBJS    := $(SOURCES:%=$(OBJDIR)/%.o)

all: $(OUT)/libscenenet_hip.so

$(OBJDIR)/%.o: $(SRC)/%.hip $(SRC)/common.h include/scenenet_hip.h $(wildcard $(SRC)/*.inc) $(SRC)/conv_prep.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) $(FLAGS_$*) -c $< -o $@

$(OUT)/libscenenet_hip.so: $(OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

asm:
	@mkdir -p build/asm
	for f in $(SOURCES); do \
	  $(HIPCC) $(CXXFLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
	    $(SRC)/$$f.hip -o build/asm/$$f.s 2> build/asm/$$f.usage.txt || exit 1; done

clean:
	rm -rf build $(OUT)

.PHONY: all asm clean
