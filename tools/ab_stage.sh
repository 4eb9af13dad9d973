#!/bin/bash
# Same-box A/B of the persistent runs (executor pass 3k): ResNet-50 bs 64 with SMELTER_STAGE=0 (every layer its own launch), the default (runs at <= 112-pixel images, kept
# where the in-situ A/B of the compile finds them faster) and SMELTER_STAGE=wide (also the 14 x 14 stage); three alternating rounds, then the per-op table of the default.
for i in 1 2 3; do
  SMELTER_STAGE=0 python tools/gpu_probe.py --model r50 2>&1 | grep " bs=" | sed 's/^/layers only      /'
  python tools/gpu_probe.py --model r50 2>&1 | grep " bs=" | sed 's/^/runs (default)   /'
  SMELTER_STAGE=wide python tools/gpu_probe.py --model r50 2>&1 | grep " bs=" | sed 's/^/runs + 14x14     /'
done
# the compile's own in-situ measurement of every run (stderr note of the autotune pass)
SMELTER_STAGE=wide SMELTER_DEBUG_PLAN=1 python tools/gpu_probe.py --model r50 2>&1 | grep "\[plan\] persistent run"
