#!/bin/bash
# Same-box A/B of the previous round's binary (a git worktree built under build/r03tree) against the current tree: three alternating runs per model.
for m in ${1:-r50 mbv2 unet}; do
  for i in 1 2 3; do
    (cd build/r03tree && python tools/gpu_probe.py --model $m 2>&1 | grep " bs=" | sed 's/^/round-3 binary  /')
    python tools/gpu_probe.py --model $m 2>&1 | grep " bs=" | sed 's/^/current         /'
  done
done
