#!/bin/bash
# Build a DIAGNOSTIC library of the persistent stage kernel (per-tile stamps, -DSM_STAGE_TL on stage.hip only) next to the product one:
#   tools/stage_tl.sh                      -> build/libsmelter_hip_stagetl.so
#   tools/stage_tl.sh nomfma -DST_ABL_NOMFMA   -> build/libsmelter_hip_stagetl_nomfma.so   (ablations: ST_ABL_NOMFMA, ST_ABL_NODMA, ST_ABL_NOW, -DST_DMA_AUX='""')
# No GPU needed; on the GPU box: SMELTER_LIB_PATH=build/libsmelter_hip_stagetl.so python tools/stage_tl.py stage4
set -e
cd "$(dirname "$0")/.."
make -s -j8
FLAGS=$(cut -d' ' -f2- build/.flags)
NAME=${1:+_$1}; shift || true
/opt/rocm/bin/hipcc $FLAGS -DSM_STAGE_TL "$@" -c smelter_amd/csrc/kernels/stage.hip -o build/stage_tl$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmelter_hip_stagetl$NAME.so build/stage_tl$NAME.o $(find build/obj -name '*.o' ! -name stage.o)
