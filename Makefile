# Builds libspice_amd.so (gfx950 HIP kernels + C ABI) in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
SRC := $(wildcard spice_amd/csrc/*.hip)
OBJ := $(patsubst spice_amd/csrc/%.hip,build/%.o,$(SRC))
LIB := spice_amd/lib/libspice_amd.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off

all: $(LIB)

build/%.o: spice_amd/csrc/%.hip spice_amd/csrc/common.h include/spice_amd.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p spice_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ) -lhipblaslt

clean:
	rm -rf build $(LIB)

.PHONY: all clean
