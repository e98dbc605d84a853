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

# header dependencies per translation unit (-MMD): editing one GEMM generation's header rebuilds only the units that include it
build/%.o: geo4d_amd/csrc/%.hip
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -MMD -MP -MF build/$*.d -c $< -o $@

-include $(OBJ:.o=.d)

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJ) -o $@

clean:
	rm -rf build $(LIB)
.PHONY: all clean
