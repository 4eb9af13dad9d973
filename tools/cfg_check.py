"""scratch: every tile config (with and without split-K workspace) against the generic kernel on odd shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smelter_amd import ops

shapes = [(2, 2, 2, 512, 512, 3, 1, 1), (2, 2, 2, 2048, 512, 1, 1, 0), (2, 4, 4, 256, 256, 3, 1, 1), (2, 4, 4, 1024, 256, 1, 1, 0),
          (2, 1, 1, 2048, 1000, 1, 1, 0), (2, 8, 8, 128, 128, 3, 1, 1), (2, 16, 16, 64, 64, 3, 1, 1), (2, 16, 16, 64, 256, 1, 1, 0),
          (2, 4, 4, 512, 1024, 1, 2, 0), (3, 7, 5, 192, 72, 3, 2, 1)]
torch.manual_seed(0)
for (n, h, w, cin, cout, k, s, p) in shapes:
    x = torch.randn((n, h, w, cin), device="cuda").half()
    wt = (torch.randn((cout, k, k, cin), device="cuda") * (2.0 / (cin * k * k)) ** 0.5).half()
    b = torch.randn(cout, device="cuda")
    ho, wo = ops.conv_out_size(h, w, k, k, s, s, (p, p, p, p))
    r = torch.randn((n, ho, wo, cout), device="cuda").half()
    ref = ops.conv2d(x, wt, b, residual=r, stride=(s, s), pads=(p, p, p, p), act=1, cfg=3).float().cpu().numpy()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for cfg in range(ops.num_conv_configs()):
        for use_ws in (False, True):
            try:
                y = ops.conv2d(x, wt, b, residual=r, stride=(s, s), pads=(p, p, p, p), act=1, cfg=cfg, workspace=ws if use_ws else None)
            except Exception as e:
                continue
            y = y.float().cpu().numpy()
            err = np.abs(y - ref).max() / np.abs(ref).max()
            if not np.isfinite(y).all() or err > 2e-3:
                print("BAD shape", (n, h, w, cin, cout, k, s, p), "cfg", cfg, "ws", use_ws, "err", err, "finite", np.isfinite(y).all())
print("done")
