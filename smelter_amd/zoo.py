"""Synthetic ONNX models for the BASELINE.json configs (seeded random weights, no downloads).

Recipe = SURVEY.md §8(d): He-normal conv weights (rng seed 1), BN gamma U[0.5,1.5], beta N(0,0.1),
mean N(0,0.1), var U[0.5,1.5], eps 1e-5, FC N(0,0.01).  Every op used is in the reference's
converter registry (/root/reference/Sources/Smelter/ONNXGraph.swift:110-155): Conv,
BatchNormalization, Relu, Add, MaxPool, GlobalAveragePool, Flatten, Gemm, Upsample, Concat.

The last BatchNormalization of every residual branch gets gamma scaled by `res_gamma` (default 0.25,
the usual "small residual gamma" initialisation) so that activations of a 16-block random ResNet stay
well inside fp16 range; this is a property of the synthetic weights, not of the executor.
"""
from __future__ import annotations

import numpy as np

from . import onnx_writer as ow


class _Net:
    def __init__(self, name: str, seed: int, dtype=np.float32):
        self.g = ow.GraphBuilder(name)
        self.rng = np.random.default_rng(seed)
        self.dtype = np.dtype(dtype)
        self.macs = 0  # conv + fc multiply-accumulates per image

    # -- parameter factories ---------------------------------------------------------------
    def conv_w(self, cout, cin_g, kh, kw):
        std = np.sqrt(2.0 / (cin_g * kh * kw))
        return (self.rng.standard_normal((cout, cin_g, kh, kw)) * std).astype(self.dtype)

    def bn_params(self, c, gamma_scale=1.0):
        r = self.rng
        gamma = (r.uniform(0.5, 1.5, c) * gamma_scale).astype(self.dtype)
        beta = (r.standard_normal(c) * 0.1).astype(self.dtype)
        mean = (r.standard_normal(c) * 0.1).astype(self.dtype)
        var = r.uniform(0.5, 1.5, c).astype(self.dtype)
        return gamma, beta, mean, var

    # -- layers ------------------------------------------------------------------------------
    def conv(self, x, hw, cin, cout, k, s=1, p=None, group=1, bias=False):
        p = k // 2 if p is None else p
        g = self.g
        wname = g.add_init(g.fresh("w"), self.conv_w(cout, cin // group, k, k))
        ins = [x, wname]
        if bias:
            ins.append(g.add_init(g.fresh("b"), (self.rng.standard_normal(cout) * 0.1).astype(self.dtype)))
        y = g.add_node("Conv", ins, dilations=[1, 1], group=group, kernel_shape=[k, k],
                       pads=[p, p, p, p], strides=[s, s])
        ho = (hw[0] + 2 * p - k) // s + 1
        wo = (hw[1] + 2 * p - k) // s + 1
        self.macs += ho * wo * cout * (cin // group) * k * k
        return y, (ho, wo)

    def bn(self, x, c, gamma_scale=1.0):
        g = self.g
        names = [g.add_init(g.fresh(n), a) for n, a in zip(("gamma", "beta", "mean", "var"),
                                                            self.bn_params(c, gamma_scale))]
        return g.add_node("BatchNormalization", [x] + names, epsilon=1e-5, momentum=0.9)

    relu6 = None                             # None: Relu nodes; "attr" / "input": every activation is ONNX Clip(0, 6) in its opset-6 / opset-11 form

    def relu(self, x):
        if self.relu6 == "attr":
            return self.g.add_node("Clip", [x], min=0.0, max=6.0)
        if self.relu6 == "input":
            lo = self.g.add_init(self.g.fresh("clip_min"), np.zeros((), np.float32))
            hi = self.g.add_init(self.g.fresh("clip_max"), np.full((), 6.0, np.float32))
            return self.g.add_node("Clip", [x, lo, hi])
        return self.g.add_node("Relu", [x])

    def conv_bn(self, x, hw, cin, cout, k, s=1, group=1, act=True, gamma_scale=1.0, p=None):
        y, hw = self.conv(x, hw, cin, cout, k, s, p=p, group=group)
        y = self.bn(y, cout, gamma_scale)
        if act:
            y = self.relu(y)
        return y, hw

    def maxpool(self, x, hw, k, s, p):
        y = self.g.add_node("MaxPool", [x], kernel_shape=[k, k], pads=[p, p, p, p], strides=[s, s])
        return y, ((hw[0] + 2 * p - k) // s + 1, (hw[1] + 2 * p - k) // s + 1)

    def head(self, x, cin, classes):
        g = self.g
        y = g.add_node("GlobalAveragePool", [x])
        y = g.add_node("Flatten", [y], axis=1)
        w = g.add_init(g.fresh("fc_w"), (self.rng.standard_normal((classes, cin)) * 0.01).astype(self.dtype))
        b = g.add_init(g.fresh("fc_b"), (self.rng.standard_normal(classes) * 0.01).astype(self.dtype))
        self.macs += cin * classes
        return g.add_node("Gemm", [y, w, b], alpha=1.0, beta=1.0, transB=1)


def _finish(net: _Net, out, in_dims, out_dims, **kw):
    elem = ow.FLOAT16 if net.dtype == np.float16 else ow.FLOAT
    net.g.inputs.insert(0, ow.value_info("input", elem, in_dims))
    net.g.add_output(out, out_dims, elem)
    return net.g.to_model(**kw)


def toy(seed: int = 1, hw=(64, 64), dtype=np.float32, with_bn: bool = True) -> bytes:
    """Config 1: 3 x [Conv3x3 p1 (3->8->16->16) -> BN -> ReLU], input 1x3x64x64."""
    net = _Net("toy", seed, dtype)
    x, cur = "input", hw
    for cin, cout in ((3, 8), (8, 16), (16, 16)):
        if with_bn:
            x, cur = net.conv_bn(x, cur, cin, cout, 3)
        else:
            x, cur = net.conv(x, cur, cin, cout, 3, bias=True)
            x = net.relu(x)
    return _finish(net, x, [1, 3, hw[0], hw[1]], [1, 16, cur[0], cur[1]])


def resnet50(seed: int = 1, batch: int = 1, hw=(224, 224), classes: int = 1000, res_gamma: float = 0.25,
             dtype=np.float32, layers=(3, 4, 6, 3), width: int = 64, return_macs: bool = False, shortcut_order: str = "last"):
    """Config 2/5: ResNet-50 v1.5 (stride on the 3x3), 53 Conv + 53 BN + 49 Relu + 16 Add + MaxPool +
    GlobalAveragePool + Flatten + Gemm = 175 nodes."""
    net = _Net("resnet50", seed, dtype)
    x, cur = net.conv_bn("input", hw, 3, width, 7, s=2, p=3)
    x, cur = net.maxpool(x, cur, 3, 2, 1)
    cin = width
    for stage, nblocks in enumerate(layers):
        mid = width << stage
        cout = mid * 4
        for b in range(nblocks):
            s = 2 if (b == 0 and stage > 0) else 1
            identity = x
            # node order of the projection shortcut relative to the main branch: "last" = the usual export (the shortcut convolution runs last and
            # carries the fused Add), "mid" = between the 3x3 and the expand layer, "first" = in front of the block (the expand layer carries the Add)
            if b == 0 and shortcut_order == "first":
                identity, _ = net.conv_bn(x, cur, cin, cout, 1, s=s, act=False)
            y, hw1 = net.conv_bn(x, cur, cin, mid, 1)
            y, hw2 = net.conv_bn(y, hw1, mid, mid, 3, s=s)
            if b == 0 and shortcut_order == "mid":
                identity, _ = net.conv_bn(x, cur, cin, cout, 1, s=s, act=False)
            y, hw3 = net.conv_bn(y, hw2, mid, cout, 1, act=False, gamma_scale=res_gamma)
            if b == 0 and shortcut_order == "last":
                identity, _ = net.conv_bn(x, cur, cin, cout, 1, s=s, act=False)
            x = net.relu(net.g.add_node("Add", [y, identity]))
            cur, cin = hw3, cout
    out = net.head(x, cin, classes)
    m = _finish(net, out, [batch, 3, hw[0], hw[1]], [batch, classes])
    return (m, net.macs) if return_macs else m


_MBV2 = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def mobilenet_v2(seed: int = 1, batch: int = 1, hw=(224, 224), classes: int = 1000, res_gamma: float = 0.5,
                 dtype=np.float32, return_macs: bool = False, relu6: str = None):
    """Config 3: MobileNet-v2 1.0, 52 Conv + 1 Gemm.  By default Relu stands in for ReLU6: `Clip` is not in the
    reference registry (ONNXGraph.swift:110-155), see SURVEY.md §2a.  relu6 = "attr" | "input": the real thing, every activation an ONNX
    Clip(0, 6) with the bounds as attributes (opset 6) or as constant inputs (opset 11, what a torchvision export holds) — the
    declared-superset ClipConverter (converters.cpp) takes both."""
    net = _Net("mobilenet_v2", seed, dtype)
    net.relu6 = relu6
    x, cur = net.conv_bn("input", hw, 3, 32, 3, s=2)
    cin = 32
    for t, c, n, s in _MBV2:
        for i in range(n):
            stride = s if i == 0 else 1
            hidden = cin * t
            inp, y, h = x, x, cur
            if t != 1:
                y, h = net.conv_bn(y, h, cin, hidden, 1)
            y, h = net.conv_bn(y, h, hidden, hidden, 3, s=stride, group=hidden)
            use_res = stride == 1 and cin == c
            y, h = net.conv_bn(y, h, hidden, c, 1, act=False, gamma_scale=res_gamma if use_res else 1.0)
            x = net.g.add_node("Add", [y, inp]) if use_res else y
            cur, cin = h, c
    x, cur = net.conv_bn(x, cur, cin, 1280, 1)
    out = net.head(x, 1280, classes)
    m = _finish(net, out, [batch, 3, hw[0], hw[1]], [batch, classes])
    return (m, net.macs) if return_macs else m


def unet(seed: int = 1, batch: int = 1, hw=(512, 512), base: int = 32, levels: int = 4, out_ch: int = 8,
         mode: str = "nearest", dtype=np.float32, with_bn: bool = False, return_macs: bool = False):
    """Config 4: encoder/decoder with Upsample + Concat skips: per level [Conv3x3-ReLU]x2,
    MaxPool 2x2 s2; decoder = Upsample x2 -> Conv3x3 (C -> C/2) -> Concat(skip) -> [Conv3x3-ReLU]x2;
    1x1 head.  levels=4, base=32 -> 23 convs, 8.63 M weights.  Every Concat has exactly two operands of
    equal C (C % 4 == 0), the only case the reference's ConcatConverter records correctly
    (Converters.swift:566-567, README.md:63)."""
    net = _Net("unet", seed, dtype)

    def block(x, cur, cin, cout):
        for ci in (cin, cout):
            if with_bn:
                x, cur = net.conv_bn(x, cur, ci, cout, 3)
            else:
                x, cur = net.conv(x, cur, ci, cout, 3, bias=True)
                x = net.relu(x)
        return x, cur

    x, cur, cin = "input", hw, 3
    skips = []
    for lv in range(levels):
        c = base << lv
        x, cur = block(x, cur, cin, c)
        skips.append((x, c, cur))
        x, cur = net.maxpool(x, cur, 2, 2, 0)
        cin = c
    x, cur = block(x, cur, cin, base << levels)
    cin = base << levels
    for lv in reversed(range(levels)):
        skip, c, shw = skips[lv]
        x = net.g.add_node("Upsample", [x], mode=mode, scales=[1.0, 1.0, 2.0, 2.0])
        cur = (cur[0] * 2, cur[1] * 2)
        x, cur = net.conv(x, cur, cin, c, 3, bias=True)
        x = net.g.add_node("Concat", [x, skip], axis=1)
        x, cur = block(x, cur, 2 * c, c)
        cin = c
    x, cur = net.conv(x, cur, cin, out_ch, 1, p=0, bias=True)
    m = _finish(net, x, [batch, 3, hw[0], hw[1]], [batch, out_ch, cur[0], cur[1]])
    return (m, net.macs) if return_macs else m


def deconv_net(seed: int = 1, batch: int = 1, hw=(32, 32), base: int = 16, dtype=np.float32, return_macs: bool = False):
    """a small segmentation-style decoder built on ConvTranspose instead of Upsample:
    Conv3x3(3->base)-ReLU, MaxPool2, Conv3x3(base->2base)-BN-ReLU, MaxPool2, ConvTranspose 2x2 s2 (2base->base)-BN-ReLU,
    Concat(skip), ConvTranspose 3x3 s2 p1 output_padding 1 (2base->base)-ReLU, Conv1x1 head."""
    net = _Net("deconv_net", seed, dtype)
    g = net.g

    def convT(x, cur, cin, cout, k, s, p, op, bias=True):
        std = np.sqrt(2.0 / (cin * k * k / (s * s)))
        w = g.add_init(g.fresh("wt"), (net.rng.standard_normal((cin, cout, k, k)) * std).astype(net.dtype))
        ins = [x, w]
        if bias:
            ins.append(g.add_init(g.fresh("b"), (net.rng.standard_normal(cout) * 0.1).astype(net.dtype)))
        y = g.add_node("ConvTranspose", ins, dilations=[1, 1], group=1, kernel_shape=[k, k], pads=[p, p, p, p], strides=[s, s],
                       output_padding=[op, op])
        net.macs += cur[0] * cur[1] * cin * cout * k * k
        return y, ((cur[0] - 1) * s - 2 * p + k + op, (cur[1] - 1) * s - 2 * p + k + op)

    x, cur = net.conv("input", hw, 3, base, 3, bias=True)
    x = net.relu(x)
    skip, skip_hw = x, cur
    x, cur = net.maxpool(x, cur, 2, 2, 0)
    x, cur = net.conv_bn(x, cur, base, 2 * base, 3)
    x, cur = net.maxpool(x, cur, 2, 2, 0)
    x, cur = convT(x, cur, 2 * base, base, 2, 2, 0, 0, bias=False)
    x = net.relu(net.bn(x, base))
    x2, cur2 = net.maxpool(skip, skip_hw, 2, 2, 0)
    assert cur2 == cur
    x = g.add_node("Concat", [x, x2], axis=1)
    x, cur = convT(x, cur, 2 * base, base, 3, 2, 1, 1)
    x = net.relu(x)
    x, cur = net.conv(x, cur, base, 8, 1, p=0, bias=True)
    m = _finish(net, x, [batch, 3, hw[0], hw[1]], [batch, 8, cur[0], cur[1]])
    return (m, net.macs) if return_macs else m


def misc_ops_net(seed: int = 1, batch: int = 1, hw=(24, 20), final: str = "LogSoftmax", dtype=np.float32):
    """one graph through every remaining converter of ONNXGraph.swift:110-155 — Sub/Mul/Div, Sigmoid/HardSigmoid/
    Tanh/Elu/Abs/Softplus/Softsign/Exp/Log/Pow, Pad (reflect, edge, constant), InstanceNormalization, custom_group_norm, Softmax/LogSoftmax."""
    net = _Net("misc_ops", seed, dtype)
    g = net.g
    # the two normalisation layers sit right behind convolutions of well-spread data (1/sigma amplifies fp16 rounding noise)
    x, cur = net.conv("input", hw, 3, 16, 3, bias=True)
    gam = g.add_init(g.fresh("in_g"), net.rng.uniform(0.5, 1.5, 16).astype(net.dtype))
    bet = g.add_init(g.fresh("in_b"), (net.rng.standard_normal(16) * 0.1).astype(net.dtype))
    n1 = g.add_node("InstanceNormalization", [x, gam, bet], epsilon=1e-5)
    a = g.add_node("Sigmoid", [n1])
    b, _ = net.conv(a, cur, 16, 16, 1, p=0, bias=True)
    b = g.add_node("Tanh", [b])                                      # fused into the conv epilogue
    m = g.add_node("Mul", [a, b])
    sp = g.add_node("Softplus", [b])                                 # > 0: a safe divisor
    d = g.add_node("Div", [m, sp])
    s = g.add_node("Sub", [d, a])
    e = g.add_node("Elu", [s], alpha=0.8)
    e = g.add_node("Pad", [e], pads=[0, 0, 1, 2, 0, 0, 2, 1], mode=b"reflect")
    cur = (cur[0] + 3, cur[1] + 3)
    c2, cur = net.conv(e, cur, 16, 16, 3, p=0, bias=True)
    grp = g.add_init(g.fresh("gn_groups"), np.array([4], np.int64))
    gam2 = g.add_init(g.fresh("gn_g"), net.rng.uniform(0.5, 1.5, 16).astype(net.dtype))
    bet2 = g.add_init(g.fresh("gn_b"), (net.rng.standard_normal(16) * 0.1).astype(net.dtype))
    gn = g.add_node("custom_group_norm", [c2, grp, gam2, bet2])
    h1 = g.add_node("HardSigmoid", [gn])
    ab = g.add_node("Abs", [gn])
    ss = g.add_node("Softsign", [ab])
    ex = g.add_node("Exp", [ss])                                     # in [1, e)
    lg = g.add_node("Log", [ex])
    pw = g.add_node("Pow", [lg, g.add_init(g.fresh("pow_e"), np.array([2.0], net.dtype))])   # the reference ignores the exponent (identity)
    ad = g.add_node("Add", [pw, h1])
    ad = g.add_node("Pad", [ad], pads=[0, 0, 1, 1, 0, 0, 0, 0], mode=b"edge")
    gn = g.add_node("Pad", [ad], pads=[0, 0, 0, 0, 0, 0, 1, 1])      # default mode: constant 0
    cur = (cur[0] + 2, cur[1] + 2)
    y, cur = net.conv(gn, cur, 16, 8, 1, p=0, bias=True)
    y = g.add_node(final, [y], axis=1)
    return _finish(net, y, [batch, 3, hw[0], hw[1]], [batch, 8, cur[0], cur[1]])


def reshape_net(seed: int = 1, batch: int = 1, hw=(16, 12), final: str = "matmul", dtype=np.float32):
    """The converters no BASELINE model instantiates: Reshape (the reference's (W,H,C)-ordered shape with its `0` and `-1` rules,
    Converters.swift:830-876), Constant feeding Reshape and Upsample (:716-727, :505-514 scales-from-tensor), AveragePool (:607-650),
    Dropout (:918-939, two output names), plus the declared supersets LeakyRelu and MatMul.  Every kind of reshape the executor
    distinguishes occurs: spatial -> spatial (materialised), 1x1 -> 1x1 (alias), 1x1xC -> spatial (materialised), spatial -> flat
    feeding a fully-connected layer (weight columns permuted).  final = "matmul" | "reshape" (Reshape as the graph's last node)."""
    assert hw[0] % 4 == 0 and hw[1] % 4 == 0
    net = _Net("reshape_net", seed, dtype)
    g = net.g
    x, cur = net.conv("input", hw, 3, 8, 3, bias=True)
    x = g.add_node("LeakyRelu", [x], alpha=0.1)                       # fused into the conv epilogue
    x = g.add_node("Dropout", [x], n_out=2, ratio=0.5)[0]             # (output, mask): identity at inference
    x = g.add_node("AveragePool", [x], kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2])
    cur = ((cur[0] + 2 - 3) // 2 + 1, (cur[1] + 2 - 3) // 2 + 1)
    sc = g.add_node("Constant", [], value=np.array([1.0, 1.0, 2.0, 2.0], np.float32))
    x = g.add_node("Upsample", [x, sc], mode=b"nearest")              # scales from input[1]: a Constant's tensor
    cur = (cur[0] * 2, cur[1] * 2)
    x = g.add_node("AveragePool", [x], kernel_shape=[2, 2], pads=[0, 0, 0, 0], strides=[2, 2])
    cur = (cur[0] // 2, cur[1] // 2)
    total = 8 * cur[0] * cur[1]
    # (W, H, C) = (2*W, H/2, -1): spatial -> spatial, one dimension inferred
    s1 = g.add_init(g.fresh("shape"), np.array([cur[1] * 2, cur[0] // 2, -1], np.int64))
    x = g.add_node("Reshape", [x, s1])
    cur = (cur[0] // 2, cur[1] * 2)
    # `0` copies toArray()[i] of the SOURCE shape = (channels, width, height, depth)[0] = channels = 1 -> W = 1, H inferred
    s2 = g.add_node("Constant", [], value=np.array([0, -1, 8], np.int64))
    x = g.add_node("Reshape", [x, s2])
    cur = (total // 8, 1)
    x, cur = net.conv(x, cur, 8, 8, 1, p=0, bias=True)
    x = g.add_node("GlobalAveragePool", [x])
    s3 = g.add_init(g.fresh("shape"), np.array([1, 1, -1], np.int64))
    x = g.add_node("Reshape", [x, s3])                                # 1x1x8 -> 1x1x8: alias
    s4 = g.add_init(g.fresh("shape"), np.array([2, 2, 2], np.int64))
    x = g.add_node("Reshape", [x, s4])                                # 1x1x8 -> 2x2 map of 2 channels: materialised
    if final == "reshape":
        return _finish(net, x, [batch, 3, hw[0], hw[1]], [batch, 2, 2, 2])
    x = g.add_node("Sigmoid", [x])                                    # a non-FC consumer of the reshaped map
    x = g.add_node("Flatten", [x], axis=1)
    w = g.add_init(g.fresh("mm_w"), (net.rng.standard_normal((8, 5)) * 0.5).astype(net.dtype))   # MatMul: [K, N]
    net.macs += 8 * 5
    x = g.add_node("MatMul", [x, w])
    return _finish(net, x, [batch, 3, hw[0], hw[1]], [batch, 5])


def synthetic_images(batch: int, c: int = 3, hw=(224, 224), seed: int = 0) -> np.ndarray:
    """NCHW fp32 i.i.d. uniform [0,1) — what an RGBA8-unorm texture yields in the reference flow
    (/root/reference/README.md:33-39)."""
    return np.random.default_rng(seed).random((batch, c, hw[0], hw[1]), dtype=np.float32)
