"""Run sm_stage_fwd a few times on a ResNet-50 bs-64 stage shape and nothing else (the process tools/pmc_stage.sh profiles).  Usage: python tools/stage_run.py [stage4|stage3] [reps]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from stage_bench import build

which = sys.argv[1] if len(sys.argv) > 1 else "stage4"; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
hw, cin, mid, blocks = (7, 2048, 512, 2) if which == "stage4" else (14, 1024, 256, 5)
st, _ = build(64, hw, cin, mid, blocks)
for _ in range(reps):
    st.run()
torch.cuda.synchronize()
print("status", st.status(), "kinds", st.kinds())
