#!/bin/bash
# Build the DIAGNOSTIC library of the wave-split 3x3 kernel (per-phase stamps, -DSM_KQ_TL on conv_kq.hip only) next to the product one:
# build/libsmelter_hip_kqtl.so.  No GPU needed; run tools/kq_tl.py --lib build/libsmelter_hip_kqtl.so on the GPU box afterwards.
set -e
cd "$(dirname "$0")/.."
make -s -j8
FLAGS=$(cut -d' ' -f2- build/.flags)
/opt/rocm/bin/hipcc $FLAGS -DSM_KQ_TL -c smelter_amd/csrc/kernels/conv_kq.hip -o build/conv_kq_tl.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmelter_hip_kqtl.so build/conv_kq_tl.o $(find build/obj -name '*.o' ! -name conv_kq.o)
