#!/bin/bash
# Same-box A/B of the previous round's binary (a git worktree built under build/$PREV, default r04tree: `git worktree add build/r04tree <round-4 head> && make -C build/r04tree`)
# against the current tree: three alternating runs per model.
PREV=${PREV:-r04tree}
for m in ${1:-r50 mbv2 unet}; do
  for i in 1 2 3; do
    (cd build/$PREV && python tools/gpu_probe.py --model $m 2>&1 | grep " bs=" | sed 's/^/previous round binary  /')
    python tools/gpu_probe.py --model $m 2>&1 | grep " bs=" | sed 's/^/current                /'
  done
done
