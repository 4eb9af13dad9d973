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

# header dependencies come from the compiler (-MMD): touching one kernel header rebuilds only the translation units that include it
$(OBJDIR)/%.o: $(CSRC)/%.hip $(FLAGSTAMP)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -MMD -MP -c $< -o $@
$(OBJDIR)/%.o: $(CSRC)/%.cpp $(FLAGSTAMP)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -MMD -MP -x hip -c $< -o $@
-include $(OBJS:.o=.d)
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -s -C oracle liborc.so

# CPU-only AddressSanitizer + UBSan build of the HOST sources (ONNX decoder, graph builder, converters, planner, weight packing):
# ROCm's clang as a plain x86-64 compiler, kernel entry points stubbed (tools/asan/gen_stubs.py), never run on a GPU.
# build/asan/host_fuzz (tools/asan/host_fuzz.cpp) runs model files + structured mutations through it; tests/test_sanitizer.py drives it.
ASAN_BIN := build/asan/host_fuzz
CLANGXX  ?= /opt/rocm/lib/llvm/bin/clang++
asan: $(ASAN_BIN)
$(ASAN_BIN): $(HOSTSRC) $(wildcard $(CSRC)/host/*.hpp) $(CSRC)/kernels/common.hpp include/smelter_hip.h tools/asan/gen_stubs.py tools/asan/host_fuzz.cpp
	@mkdir -p build/asan
	python3 tools/asan/gen_stubs.py build/asan/stubs.cpp
	$(CLANGXX) -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
	    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Wno-unused-result -Wno-deprecated-declarations \
	    $(HOSTSRC) build/asan/stubs.cpp tools/asan/host_fuzz.cpp -o $@ -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib

clean:
	rm -rf build $(LIB); $(MAKE) -C oracle clean
.PHONY: all oracle clean asan FORCE
