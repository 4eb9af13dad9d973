"""Is the admission of the persistent runs (executor pass 3k: ONE launch against its layers, decided by the compile's in-situ A/B) the same from compile to compile?
    python tools/run_admission.py [--model r50|mbv2|unet] [--compiles 5] [--batch N]
Compiles the model `--compiles` times in ONE process on one box (a fresh ONNXGraph + executor each time: autotune, A/B, hipGraph capture) and prints, per compile, the `on`
vector with the two medians behind every decision, the number of launches of a step and the step time; the last line says whether the vectors agree."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from smelter_amd import zoo  # noqa: E402
from smelter_amd.graph import Configuration, ONNXGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="r50")
    ap.add_argument("--compiles", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0)
    a = ap.parse_args()
    make, hw, bs = {"r50": (zoo.resnet50, (224, 224), 64), "mbv2": (zoo.mobilenet_v2, (224, 224), 256), "unet": (zoo.unet, (512, 512), 16)}[a.model]
    bs = a.batch or bs
    m = make()
    x = torch.from_numpy(zoo.synthetic_images(bs, 3, hw)).cuda()
    vectors = []
    for c in range(a.compiles):
        t = time.time()
        ex = ONNXGraph(m, Configuration(autotune=True)).hipGraph(0, bs)
        dt = time.time() - t
        for _ in range(5):
            ex.encode(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ex.encode(x)
        e1.record(); torch.cuda.synchronize()
        runs = ex.persistentRuns
        on = [int(r["on"]) for r in runs]
        vectors.append(on)
        print("compile %d (%.1f s): on = %s  launches %d  step %.4f ms" % (c, dt, on, ex.numLaunches, e0.elapsed_time(e1) / 50))
        for r in runs:
            print("    run at step %d, %d layers: one launch %.1f us | layer by layer %.1f us (medians of %d)  mode bits %d  first error %d  %s" % (
                r["first_step"], r["layers"], r["t_run_us"], r["t_layers_us"], r["samples"], r["mode_bits"], r["first_err"], "kept" if r["on"] else "turned down: " + ", ".join(r["why_off"])))
        del ex
        torch.cuda.empty_cache()
    print("same `on` vector in all %d compiles: %s" % (a.compiles, "yes" if all(v == vectors[0] for v in vectors) else "NO"))


if __name__ == "__main__":
    main()
