#!/bin/bash
# Build the DIAGNOSTIC libraries of the whole-block kernels (bneck.hip, bneck128.hpp) next to the product one, linked against the product objects.  No GPU needed.
#   build/libsmelter_hip_bntl.so   -DSM_BN_TL: per-wave s_memtime stamps (tools/bneck_tl.py at 56 x 56, tools/bneck128_tl.py at 28 x 28)
#   build/libsmelter_hip_bnabl.so  -DSM_ABLATION: the SM_BN_DEBUG environment bits (1 no stores, 2 no residual, 4 no x loads, 8 no 3x3 work, 16 no expand MFMAs, 32 no sliding window)
# Run with SMELTER_LIB_PATH=build/libsmelter_hip_bntl.so python tools/bneck_tl.py; delete the libraries afterwards (build/ ships with every gpurun call).
set -e
cd "$(dirname "$0")/.."
make -s -j8
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result"
OBJS=$(ls build/obj/kernels/*.o build/obj/host/*.o | grep -v bneck.o)
/opt/rocm/bin/hipcc -DSM_BN_TL $F -c smelter_amd/csrc/kernels/bneck.hip -o build/bneck_tl.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmelter_hip_bntl.so $OBJS build/bneck_tl.o
if [ "$1" = "abl" ]; then
  /opt/rocm/bin/hipcc -DSM_ABLATION $F -c smelter_amd/csrc/kernels/bneck.hip -o build/bneck_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmelter_hip_bnabl.so $OBJS build/bneck_abl.o
fi
