"""scratch: layer-by-layer parity diagnosis on the GPU box"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa

import oracle as O  # noqa
from smelter_amd import onnx2mps, zoo  # noqa
from smelter_amd.graph import Configuration, ONNXGraph  # noqa


def nerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def layerwise(name, model, x):
    ex = ONNXGraph(model, Configuration(useHipGraph=False, keepIntermediates=True)).hipGraph(0, x.shape[0])
    ex.encode(torch.from_numpy(x).cuda())
    got_final = ex.toFloatArray()
    og = O.Graph(model)
    og.set_round_list(ex.valueNames)
    ref_final = og.run(x)
    print(name, "final nerr", nerr(got_final, ref_final), "ops", ex.numOps)
    worst = []
    for i, vn in enumerate(ex.valueNames[1:]):
        try:
            got = ex.readValue(vn)
        except Exception as e:
            print("  readValue failed", vn, e)
            continue
        v = og.value(vn)
        if v is None:
            print("  oracle has no", vn)
            continue
        p, c, h, w = v
        ref = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (x.shape[0] * c * h * w,)).reshape(x.shape[0], c, h, w)
        got = got.reshape(ref.shape)
        worst.append((nerr(got, ref), vn, ref.shape, float(np.abs(ref).max())))
    for e, vn, sh, mx in worst:
        if e > 3e-4:
            print("   %.2e %s %s max|ref|=%.3g" % (e, vn, sh, mx))
    for i in range(ex.numOps):
        print("   op", i, ex.opDesc(i))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "mbv2"
    if which == "misc":
        layerwise("misc", onnx2mps.optimize_model(zoo.misc_ops_net(final="Softmax"), half=True), zoo.synthetic_images(2, 3, (24, 20)))
    elif which == "toy":
        layerwise("toy", zoo.toy(), zoo.synthetic_images(2, 3, (64, 64)))
    elif which == "mbv2":
        layerwise("mbv2", onnx2mps.optimize_model(zoo.mobilenet_v2(hw=(64, 64)), half=True), zoo.synthetic_images(3, 3, (64, 64)))
    elif which == "unet":
        layerwise("unet", onnx2mps.optimize_model(zoo.unet(hw=(64, 48)), half=True), zoo.synthetic_images(2, 3, (64, 48)))
    elif which == "r50":
        layerwise("r50", onnx2mps.optimize_model(zoo.resnet50(hw=(64, 64)), half=True), zoo.synthetic_images(1, 3, (64, 64)))
