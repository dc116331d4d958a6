# Builds, in-tree:
#   spice_amd/lib/libspice_amd.so    gfx950 HIP kernels + the C ABI of include/spice_amd.h
#   spice_amd/lib/libspice_torch.so  torch.ops.spice.* (TORCH_LIBRARY) over that C ABI; host C++ only
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PYTHON ?= python3
SRC := $(wildcard spice_amd/csrc/*.hip)
OBJ := $(patsubst spice_amd/csrc/%.hip,build/%.o,$(SRC))
LIB := spice_amd/lib/libspice_amd.so
TORCH_LIB := spice_amd/lib/libspice_torch.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off
TORCH_DIR := $(shell $(PYTHON) -c "import os, torch; print(os.path.dirname(torch.__file__))")
TORCH_CXXFLAGS := -O2 -fPIC -std=c++17 -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 \
	-D_GLIBCXX_USE_CXX11_ABI=1 -isystem $(TORCH_DIR)/include -isystem $(TORCH_DIR)/include/torch/csrc/api/include \
	-isystem /opt/rocm/include

all: $(LIB) $(TORCH_LIB)

build/%.o: spice_amd/csrc/%.hip $(wildcard spice_amd/csrc/*.h) include/spice_amd.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p spice_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ) -lhipblaslt

build/torch_ext.o: spice_amd/csrc/torch_ext.cpp include/spice_amd.h
	@mkdir -p build
	$(HIPCC) $(TORCH_CXXFLAGS) -c $< -o $@

$(TORCH_LIB): build/torch_ext.o $(LIB)
	$(HIPCC) -shared -fPIC -o $@ build/torch_ext.o -L$(TORCH_DIR)/lib -lc10 -lc10_hip -ltorch_cpu -ltorch \
		-Lspice_amd/lib -lspice_amd -Wl,-rpath,'$$ORIGIN'

clean:
	rm -rf build $(LIB) $(TORCH_LIB)

.PHONY: all clean
