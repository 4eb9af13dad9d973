#!/bin/bash
# conv1x1_rw.hpp against the tiled kernels on the ResNet-50 layer shapes it targets, cold caches (SM_TIME_COLD: 768 MB memset before every launch)
export SM_TIME_COLD=1
for sh in "64,14,14,256,1024,1,1,0 --res" "64,28,28,128,512,1,1,0 --res" "64,56,56,256,64,1,1,0" "64,56,56,256,128,1,1,0" "64,56,56,64,64,1,1,0" "64,56,56,64,256,1,1,0 --res"; do
  echo "== shape $sh"
  python tools/conv_bench.py --shape $sh --iters 20 2>&1 | sort -k3 -n | head -${1:-6}
done
