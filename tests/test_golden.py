"""The C oracle against the committed golden fixtures (tests/golden/*.npz, made by tools/make_golden.py with
torch.nn.functional in fp64 in the build container).  Tolerance: the oracle accumulates in fp32, so
|oracle - golden| <= 2e-5 * max|golden| (K <= 400 here)."""
import glob
import os

import numpy as np
import pytest

import oracle as O
from smelter_amd import zoo

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def close(a, b, tol=TOL):
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_fixture_inventory():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(G, "*.npz")))
    assert len(names) >= 28 and "toy_end_to_end" in names and "ct3x3_s2_p1_op1" in names


@pytest.mark.parametrize("name", ["c1x1_s1", "c1x1_s2", "c3x3_s1", "c3x3_s2", "c7x7_s2_p3", "dw3x3_s1", "dw3x3_s2"])
def test_conv(name):
    d = load(name)
    k, s, p, g = (int(v) for v in d["meta"])
    w = d["w_oihw"]
    oc, icg = w.shape[:2]
    w_ohwi = O.reformat_conv_weight(w, oc, icg, k, k).reshape(oc, k, k, icg)
    close(O.conv2d(d["x"], w_ohwi, d["b"], stride=(s, s), pads=(p, p, p, p), groups=g), d["y"])


@pytest.mark.parametrize("name", ["ct2x2_s2", "ct3x3_s2_p1_op1", "ct4x4_s2_p1", "ct3x3_s1_p1", "ct3x3_s3_op2"])
def test_conv_transpose(name):
    d = load(name)
    k, s, p, op = (int(v) for v in d["meta"])
    w = d["w_iohw"]
    ic, oc = w.shape[:2]
    w_flipped = O.reformat_conv_weight(w, oc, ic, k, k, is_transpose=True).reshape(oc, k, k, ic)   # what ConvDataSource hands MPS
    close(O.conv_transpose2d(d["x"], w_flipped, d["b"], stride=(s, s), pads=(p, p, p, p), out_pad=(op, op)), d["y"])


def test_rank3_elementwise_and_normalisation_ops():
    d = load("r3_unary")
    for kind in ("sigmoid", "hardsigmoid", "tanh", "abs", "softplus", "softsign", "exp"):
        close(O.neuron(kind, d["x"]), d[kind])
    close(O.neuron("elu", d["x"], 0.7), d["elu"])
    close(O.neuron("log", d["log_in"]), d["log"])
    d = load("r3_binary")
    for op in ("sub", "mul", "div"):
        close(O.binary(op, d["a"], d["b"]), d[op])
    d = load("r3_softmax")
    close(O.softmax(d["x"]), d["softmax"]); close(O.softmax(d["x"], log=True), d["logsoftmax"])
    d = load("r3_norms")
    close(O.group_norm(d["x"], 6, d["gamma"], d["beta"]), d["instance"]); close(O.group_norm(d["x"], 3, d["gamma"], d["beta"]), d["group3"])
    d = load("r3_resize")     # Configuration.inputConstraint .forceInputScale(.bilinear): pixel-centre sampling, edge clamp
    close(O.resize(d["x"], (8, 9), "bilinear"), d["down"]); close(O.resize(d["x"], (40, 31), "bilinear"), d["up"])
    d = load("r3_pad")
    pads = tuple(int(v) for v in d["pads"])
    for mode in ("constant", "reflect", "edge"):
        close(O.pad2d(d["x"], pads, mode), d[mode], 0)


def test_pools_and_gap():
    d = load("maxpool_3x3_s2_p1"); close(O.maxpool(d["x"], (3, 3), (2, 2), (1, 1)), d["y"], 0)
    d = load("maxpool_2x2_s2"); close(O.maxpool(d["x"], (2, 2), (2, 2), (0, 0)), d["y"], 0)
    d = load("gap_7x7"); close(O.global_avgpool(d["x"]), d["y"])


def test_upsample():
    d = load("up_nearest_x2"); close(O.upsample(d["x"], 2, 2, False), d["y"], 0)
    d = load("up_bilinear_x2_ac1"); close(O.upsample(d["x"], 2, 2, True, True), d["y"])
    d = load("up_bilinear_x2_ac0"); close(O.upsample(d["x"], 2, 2, True, False), d["y"])
    d = load("up_bilinear_3x2_ac1"); close(O.upsample(d["x"], 3, 2, True, True), d["y"])


def test_batchnorm_and_linear():
    d = load("batchnorm"); close(O.batchnorm(d["x"], d["gamma"], d["beta"], d["mean"], d["var"], 1e-5), d["y"])
    d = load("linear"); close(O.gemm_fc(d["x"], d["w"], d["b"]), d["y"])


def test_toy_network_end_to_end():
    d = load("toy_end_to_end")
    assert np.array_equal(d["x"], zoo.synthetic_images(1, 3, (64, 64)))       # the seeded input is reproducible
    close(O.Graph(zoo.toy()).run(d["x"]), d["y"])
