# Build libdecomp_hip.so (hand-written HIP kernels + C ABI) for MI355X / gfx950.
# hipcc cross-compiles without a GPU.  The .so is built in-tree (git-ignored) so that it
# travels with the repo snapshot to the GPU box.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := decomp_amd/csrc
BUILD      := build/obj
LIBDIR     := decomp_amd/lib
LIB        := $(LIBDIR)/libdecomp_hip.so
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude
SRCS       := $(wildcard $(CSRC)/*.hip)
OBJS       := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(SRCS))
HDRS       := $(wildcard $(CSRC)/*.hpp) include/decomp_hip.h

all: $(LIB)

$(BUILD)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
