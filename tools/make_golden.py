"""Generate tests/golden/*.npz in the BUILD container (torch CPU is available there; nothing here runs on the
GPU box).  Each fixture = seeded inputs + outputs of torch.nn.functional — a third-party implementation of the
ONNX operator semantics, NOT the reference (which has no arithmetic of its own for these ops: it delegates to Apple
MPS) — in fp64 where it matters, stored as fp32.  tests/test_golden.py checks the C oracle against
these; the GPU tests then check the HIP path against the oracle and, for the toy network, against the fixture.

    python tools/make_golden.py        # rewrites tests/golden/
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smelter_amd import onnx_reader as orr  # noqa: E402
from smelter_amd import zoo  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def t64(a):
    return torch.from_numpy(np.asarray(a, np.float64))


def conv_cases():
    rng = np.random.default_rng(11)
    cases = {}
    # one instance of each distinct ResNet-50 conv class at reduced N/H/W, + depthwise
    specs = {"c1x1_s1": (2, 16, 12, 12, 24, 1, 1, 0, 1), "c1x1_s2": (2, 16, 12, 12, 24, 1, 2, 0, 1), "c3x3_s1": (2, 8, 10, 9, 16, 3, 1, 1, 1),
             "c3x3_s2": (1, 8, 11, 11, 16, 3, 2, 1, 1), "c7x7_s2_p3": (1, 3, 20, 20, 8, 7, 2, 3, 1), "dw3x3_s1": (2, 16, 9, 9, 16, 3, 1, 1, 16),
             "dw3x3_s2": (1, 16, 10, 10, 16, 3, 2, 1, 16)}
    for name, (n, cin, h, w, cout, k, s, p, g) in specs.items():
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = (rng.standard_normal((cout, cin // g, k, k)) * 0.2).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        y = F.conv2d(t64(x), t64(wt), t64(b), stride=s, padding=p, groups=g).numpy().astype(np.float32)
        cases[name] = dict(x=x, w_oihw=wt, b=b, y=y, meta=np.array([k, s, p, g]))
    return cases


def convt_cases():
    """ConvTranspose: torch.conv_transpose2d, weights [Cin, Cout, kH, kW] as ONNX stores them"""
    rng = np.random.default_rng(13)
    cases = {}
    specs = {"ct2x2_s2": (2, 8, 5, 6, 6, 2, 2, 0, 0), "ct3x3_s2_p1_op1": (2, 8, 6, 5, 4, 3, 2, 1, 1), "ct4x4_s2_p1": (1, 4, 5, 5, 8, 4, 2, 1, 0),
             "ct3x3_s1_p1": (1, 6, 7, 7, 5, 3, 1, 1, 0), "ct3x3_s3_op2": (1, 3, 4, 5, 4, 3, 3, 0, 2)}
    for name, (n, cin, h, w, cout, k, s, p, op) in specs.items():
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = (rng.standard_normal((cin, cout, k, k)) * 0.3).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        y = F.conv_transpose2d(t64(x), t64(wt), t64(b), stride=s, padding=p, output_padding=op).numpy().astype(np.float32)
        cases[name] = dict(x=x, w_iohw=wt, b=b, y=y, meta=np.array([k, s, p, op]))
    return cases


def rank3_cases():
    """elementwise / normalisation ops, torch fp64 (HardSigmoid by its ONNX formula: torch's differs)"""
    rng = np.random.default_rng(14)
    c = {}
    x = rng.standard_normal((2, 6, 5, 7)).astype(np.float32) * 2
    b = (rng.standard_normal((2, 6, 5, 7)).astype(np.float32) + 3.0)
    X = t64(x)
    c["unary"] = dict(x=x, sigmoid=torch.sigmoid(X).numpy(), hardsigmoid=torch.clamp(0.2 * X + 0.5, 0, 1).numpy(), tanh=torch.tanh(X).numpy(),
                      elu=F.elu(X, alpha=0.7).numpy(), abs=X.abs().numpy(), softplus=F.softplus(X).numpy(), softsign=F.softsign(X).numpy(),
                      exp=X.exp().numpy(), log=torch.log(X.abs() + 0.1).numpy(), log_in=(np.abs(x.astype(np.float64)) + 0.1).astype(np.float32))
    c["binary"] = dict(a=x, b=b, sub=(X - t64(b)).numpy(), mul=(X * t64(b)).numpy(), div=(X / t64(b)).numpy())
    c["softmax"] = dict(x=x, softmax=F.softmax(X, dim=1).numpy(), logsoftmax=F.log_softmax(X, dim=1).numpy())
    g, be = rng.uniform(0.5, 1.5, 6).astype(np.float32), rng.standard_normal(6).astype(np.float32)
    c["norms"] = dict(x=x, gamma=g, beta=be, instance=F.instance_norm(X, weight=t64(g), bias=t64(be), eps=1e-5).numpy(),
                      group3=F.group_norm(X, 3, t64(g), t64(be), eps=1e-5).numpy())
    xr = rng.standard_normal((2, 3, 17, 23)).astype(np.float32)
    c["resize"] = dict(x=xr, down=F.interpolate(t64(xr), size=(8, 9), mode="bilinear", align_corners=False).numpy(),
                       up=F.interpolate(t64(xr), size=(40, 31), mode="bilinear", align_corners=False).numpy())
    c["pad"] = dict(x=x, constant=F.pad(X, (2, 1, 1, 3)).numpy(), reflect=F.pad(X, (2, 1, 1, 3), mode="reflect").numpy(),
                    edge=F.pad(X, (2, 1, 1, 3), mode="replicate").numpy(), pads=np.array([1, 2, 3, 1]))   # pads = top, left, bottom, right
    return {k: {kk: np.asarray(vv, np.float32) if kk != "pads" else vv for kk, vv in v.items()} for k, v in c.items()}


def misc_cases():
    rng = np.random.default_rng(12)
    c = {}
    x = rng.standard_normal((2, 8, 12, 12)).astype(np.float32)
    # 13x13: (out-1)*s + k - in is even, the only case where the reference's window-offset rule
    # (PyTorchPoolPadding.swift:56-75, start = -extra/2) coincides with PyTorch's start = -p
    x13 = rng.standard_normal((2, 8, 13, 13)).astype(np.float32)
    c["maxpool_3x3_s2_p1"] = dict(x=x13, y=F.max_pool2d(t64(x13), 3, 2, 1).numpy().astype(np.float32))
    c["maxpool_2x2_s2"] = dict(x=x, y=F.max_pool2d(t64(x), 2, 2, 0).numpy().astype(np.float32))
    x7 = rng.standard_normal((2, 16, 7, 7)).astype(np.float32)
    c["gap_7x7"] = dict(x=x7, y=F.adaptive_avg_pool2d(t64(x7), 1).numpy().astype(np.float32))
    xs = rng.standard_normal((2, 4, 5, 6)).astype(np.float32)
    c["up_nearest_x2"] = dict(x=xs, y=F.interpolate(t64(xs), scale_factor=2, mode="nearest").numpy().astype(np.float32))
    c["up_bilinear_x2_ac1"] = dict(x=xs, y=F.interpolate(t64(xs), scale_factor=2, mode="bilinear", align_corners=True).numpy().astype(np.float32))
    c["up_bilinear_x2_ac0"] = dict(x=xs, y=F.interpolate(t64(xs), scale_factor=2, mode="bilinear", align_corners=False).numpy().astype(np.float32))
    c["up_bilinear_3x2_ac1"] = dict(x=xs, y=F.interpolate(t64(xs), size=(15, 12), mode="bilinear", align_corners=True).numpy().astype(np.float32))
    g, b, m = (rng.standard_normal(8).astype(np.float32) for _ in range(3))
    v = rng.uniform(0.5, 1.5, 8).astype(np.float32)
    c["batchnorm"] = dict(x=x, gamma=g, beta=b, mean=m, var=v,
                          y=F.batch_norm(t64(x), t64(m), t64(v), t64(g), t64(b), False, 0.0, 1e-5).numpy().astype(np.float32))
    xf = rng.standard_normal((3, 40)).astype(np.float32)
    wf = rng.standard_normal((10, 40)).astype(np.float32)
    bf = rng.standard_normal(10).astype(np.float32)
    c["linear"] = dict(x=xf, w=wf, b=bf, y=F.linear(t64(xf), t64(wf), t64(bf)).numpy().astype(np.float32))
    return c


def toy_end_to_end():
    """config 1 (toy Conv->BN->ReLU x3, 1x3x64x64) evaluated layer by layer with torch in fp64"""
    data = zoo.toy()
    m = orr.parse_model(data)
    arr = {t.name: t.to_array() for t in m.initializer}
    x = zoo.synthetic_images(1, 3, (64, 64))
    vals = {"input": t64(x)}
    for n in m.node:
        if n.op_type == "Conv":
            vals[n.output[0]] = F.conv2d(vals[n.input[0]], t64(arr[n.input[1]]), None, stride=n.attr("strides").ints[0], padding=n.attr("pads").ints[0])
        elif n.op_type == "BatchNormalization":
            g, b, mu, var = (t64(arr[i]) for i in n.input[1:5])
            vals[n.output[0]] = F.batch_norm(vals[n.input[0]], mu, var, g, b, False, 0.0, n.attr("epsilon").f)
        elif n.op_type == "Relu":
            vals[n.output[0]] = F.relu(vals[n.input[0]])
    y = vals[m.output[0].name].numpy().astype(np.float32)
    return dict(x=x, y=y, model_seed=np.array([1]))


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, d in {**conv_cases(), **misc_cases()}.items():
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    for name, d in convt_cases().items():
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    for name, d in rank3_cases().items():
        np.savez_compressed(os.path.join(OUT, "r3_" + name + ".npz"), **d)
    np.savez_compressed(os.path.join(OUT, "toy_end_to_end.npz"), **toy_end_to_end())
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote %d fixtures, %d KB" % (len(os.listdir(OUT)), tot // 1024))


if __name__ == "__main__":
    main()
