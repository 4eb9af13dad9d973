"""Per-launch fixed cost of a workload: step time at several batch sizes, least-squares fit t(bs) = t0 + slope * bs.
t0 is what the step pays regardless of the amount of work (launch boundaries, prologues to the first MFMA, epilogues, tails);
slope * bs is the work itself.  Writes one JSON object (profiles/r03_fixed_cost_fit.json).

    python tools/fixed_cost_fit.py [--model r50] [--batches 16,32,64] [--out gpurun_out/r03_fixed_cost_fit.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from smelter_amd import zoo  # noqa: E402
from smelter_amd.graph import Configuration, ONNXGraph  # noqa: E402


def step_ms(ex, x, steps=50, regions=5):
    for _ in range(10):
        ex.encode(x)
    torch.cuda.synchronize()
    out = []
    for _ in range(regions):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            ex.encode(x)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / steps)
    out.sort()
    return out[len(out) // 2], out[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="r50")
    ap.add_argument("--batches", default="16,32,64")
    ap.add_argument("--out", default="")
    ap.add_argument("--head", default="unrecorded")
    a = ap.parse_args()
    make, hw = {"r50": (zoo.resnet50, (224, 224)), "mbv2": (zoo.mobilenet_v2, (224, 224)), "unet": (zoo.unet, (512, 512))}[a.model]
    model = make()
    bss = [int(b) for b in a.batches.split(",")]
    rows = []
    for bs in bss:
        ex = ONNXGraph(model, Configuration(autotune=True)).hipGraph(0, bs)
        x = torch.from_numpy(zoo.synthetic_images(bs, 3, hw)).cuda()
        med, mn = step_ms(ex, x)
        rows.append({"batch": bs, "ms_per_step_median": round(med, 4), "ms_per_step_min": round(mn, 4), "launches": ex.numLaunches, "flops": ex.flops})
        print("bs=%d  %.4f ms (min %.4f)  %d launches" % (bs, med, mn, ex.numLaunches), file=sys.stderr)
        del ex, x
        torch.cuda.empty_cache()
    b = np.array([r["batch"] for r in rows], np.float64); t = np.array([r["ms_per_step_median"] for r in rows], np.float64)
    A = np.stack([np.ones_like(b), b], 1)
    (t0, slope), res, _, _ = np.linalg.lstsq(A, t, rcond=None)
    launches = rows[-1]["launches"]
    flops_img = rows[-1]["flops"] / rows[-1]["batch"]
    out = {"model": a.model, "head": a.head, "points": rows,
           "fit": {"form": "ms_per_step = t0 + slope * batch", "t0_ms": round(float(t0), 4), "slope_ms_per_image": round(float(slope), 5),
                   "max_abs_residual_ms": round(float(np.abs(A @ np.array([t0, slope]) - t).max()), 4)},
           "fixed_cost_us_per_launch": round(float(t0) * 1e3 / launches, 2), "launches_per_step": launches,
           "work_rate_tflops": round(flops_img / (float(slope) * 1e-3) / 1e12, 1),
           "fixed_share_at_largest_batch": round(float(t0) / rows[-1]["ms_per_step_median"], 3)}
    s = json.dumps(out, indent=1)
    print(s)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
