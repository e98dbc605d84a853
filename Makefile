# Build libgeo4d_hip.so (gfx950 only) and nothing else. `python -c "import __graft_entry__ as g; g.build()"` calls this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard geo4d_amd/csrc/*.hip)
OBJ   := $(patsubst geo4d_amd/csrc/%.hip,build/%.o,$(SRC))
LIB   := geo4d_amd/csrc/libgeo4d_hip.so
# -amdgpu-mfma-vgpr-form: gfx950 has a unified register file; keeping MFMA C/D in VGPRs removes the v_accvgpr_read/write
# traffic between the matrix pipe and the softmax / epilogue VALU code (attention loop 665 -> 551 instructions per tile).
CXXFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Igeo4d_amd/csrc -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $(if $(ABLATION),-DGEO4D_GEMM_ABLATION)

all: $(LIB)

build/%.o: geo4d_amd/csrc/%.hip geo4d_amd/csrc/common.h geo4d_amd/csrc/gemm_kernel.h geo4d_amd/csrc/gemm_kernel_v2.h geo4d_amd/csrc/gemm_kernel_v3.h include/geo4d_hip.h
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJ) -o $@

clean:
	rm -rf build $(LIB)
.PHONY: all clean
