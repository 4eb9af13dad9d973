# Top-level build: libsmelter_hip.so (product: HIP kernels + C++ host + C ABI) and oracle/liborc.so (checker).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
HIPFLAGS ?= $(EXTRA) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result
CSRC     := smelter_amd/csrc
KERNELS  := $(wildcard $(CSRC)/kernels/*.hip)
HOSTSRC  := $(wildcard $(CSRC)/host/*.cpp)
OBJDIR   := build/obj
OBJS     := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(KERNELS)) $(patsubst $(CSRC)/%.cpp,$(OBJDIR)/%.o,$(HOSTSRC))
LIB      := smelter_amd/libsmelter_hip.so

all: $(LIB) oracle

# objects are rebuilt when the flags change (e.g. make EXTRA=-DSM_ABLATION and then a plain make)
FLAGSTAMP := build/.flags
$(FLAGSTAMP): FORCE
	@mkdir -p build; echo '$(HIPCC) $(HIPFLAGS)' | cmp -s - $@ || echo '$(HIPCC) $(HIPFLAGS)' > $@
FORCE:

$(OBJDIR)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/kernels/*.hpp) include/smelter_hip.h $(FLAGSTAMP)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OBJDIR)/%.o: $(CSRC)/%.cpp $(wildcard $(CSRC)/host/*.hpp) $(wildcard $(CSRC)/kernels/*.hpp) include/smelter_hip.h $(FLAGSTAMP)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -s -C oracle liborc.so

clean:
	rm -rf build $(LIB); $(MAKE) -C oracle clean
.PHONY: all oracle clean FORCE
