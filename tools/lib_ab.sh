#!/bin/bash
# Same-box A/B of library BUILDS (e.g. build/libsmelter_hip_sc1.so = make ... EXTRA=-DSM_ST_AUX=16): three alternating runs of tools/gpu_probe.py per model.
#   bash tools/lib_ab.sh "r50 unet" smelter_amd/libsmelter_hip.so build/libsmelter_hip_sc1.so
cd "$(dirname "$0")/.."
MODELS=${1:-r50}; shift
for m in $MODELS; do
  for i in 1 2 3; do
    for lib in "$@"; do
      SMELTER_LIB_PATH=$(realpath $lib) python tools/gpu_probe.py --model $m 2>&1 | grep " bs=" | sed "s#^#$(basename $lib)  #"
    done
  done
done
