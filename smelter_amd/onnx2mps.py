"""Offline optimiser: plain ONNX -> "ONNX2MPS"-flavoured ONNX, the counterpart of /root/reference/ONNX2MPS.py
without the `onnx` package (absent from this image).

What the reference script does (ONNX2MPS.py:70-109) and what is restated here with numpy:
  1. optimizer.optimize(model, ['fuse_bn_into_conv'])  (:107) — the arithmetic lives in the un-vendored `onnx`
     package (version unpinned; `onnx.optimizer` was removed from onnx >= 1.9).  Published algorithm of that pass:
     for Conv -> BatchNormalization where the Conv output feeds only the BN:
         s = gamma / sqrt(var + eps);  W' = W * s[oc];  b' = (b - mean) * s + beta   (b = 0 if absent)
     the BN node is deleted and the Conv takes over its output name.
  2. cast EVERY initializer and the graph input/output infos to fp32 or fp16 (:24-31, :85-93; numpy astype, RNE).
  3. transpose Conv weights [O,I,H,W] -> [O,H,W,I] (:73-76, :58-60); ConvTranspose [I,O,H,W] -> [O,H,W,I] + flip H,W
     (:77-79, :61-62).
  4. producer_name = 'ONNX2MPS' (:98) — which makes ONNXGraph pick Format.mpsFlavor (ONNXGraph.swift:98-103).

CLI mirrors the reference (:112-134):  python -m smelter_amd.onnx2mps --input a.onnx --output b.onnx [--half]
"""
from __future__ import annotations

import argparse

import numpy as np

from . import onnx_reader as orr
from . import onnx_writer as ow


def fuse_bn_into_conv(m: orr.Model) -> orr.Model:
    inits = {t.name: t for t in m.initializer}
    arrays = {t.name: t.to_array() for t in m.initializer}
    consumers = {}
    for n in m.node:
        for i in n.input:
            consumers[i] = consumers.get(i, 0) + 1
    for o in m.output:
        consumers[o.name] = consumers.get(o.name, 0) + 1
    producer = {o: n for n in m.node for o in n.output}
    dead = set()
    new_inits = {}
    for bn in m.node:
        if bn.op_type != "BatchNormalization" or len(bn.input) < 5:
            continue
        conv = producer.get(bn.input[0])
        if conv is None or conv.op_type != "Conv" or consumers.get(bn.input[0], 0) != 1 or id(conv) in dead:
            continue
        if conv.input[1] not in arrays or any(p not in arrays for p in bn.input[1:5]):
            continue
        w = arrays[conv.input[1]].astype(np.float32)
        gamma, beta, mean, var = (arrays[p].astype(np.float32) for p in bn.input[1:5])
        eps_a = bn.attr("epsilon")
        eps = np.float32(eps_a.f if eps_a is not None else 1e-5)
        s = (gamma / np.sqrt(var + eps)).astype(np.float32)
        b = arrays[conv.input[2]].astype(np.float32) if len(conv.input) > 2 else np.zeros_like(mean)
        w2 = (w * s.reshape(-1, 1, 1, 1)).astype(np.float32)
        b2 = ((b - mean) * s + beta).astype(np.float32)
        wname, bname = conv.input[1], (conv.input[2] if len(conv.input) > 2 else conv.output[0] + "_fused_bias")
        new_inits[wname], new_inits[bname] = w2, b2
        arrays[wname], arrays[bname] = w2, b2
        conv.input = [conv.input[0], wname, bname]
        conv.output = list(bn.output[:1])
        producer[conv.output[0]] = conv
        dead.add(id(bn))
        for p in bn.input[1:5]:
            consumers[p] = consumers.get(p, 0) - 1
    m.node = [n for n in m.node if id(n) not in dead]
    used = {i for n in m.node for i in n.input}
    out_inits = []
    for t in m.initializer:
        if t.name not in used:
            continue
        if t.name in new_inits:
            a = new_inits.pop(t.name)
            t = orr.Tensor(name=t.name, dims=list(a.shape), data_type=ow.FLOAT, raw_data=a.astype("<f4").tobytes())
        out_inits.append(t)
    for name, a in new_inits.items():
        out_inits.append(orr.Tensor(name=name, dims=list(a.shape), data_type=ow.FLOAT, raw_data=a.astype("<f4").tobytes()))
    m.initializer = out_inits
    return m


def model_to_mps(m: orr.Model, half: bool) -> bytes:
    dt = np.float16 if half else np.float32
    elem = ow.FLOAT16 if half else ow.FLOAT
    plan, flip = {}, {}
    for n in m.node:                                   # ONNX2MPS.py:73-79
        if n.op_type == "Conv":
            plan[n.input[1]], flip[n.input[1]] = (0, 2, 3, 1), False
        elif n.op_type == "ConvTranspose":
            plan[n.input[1]], flip[n.input[1]] = (1, 2, 3, 0), True
    inits = []
    for t in m.initializer:                            # :24-31 — every initializer is cast, shape tensors included
        a = t.to_array().astype(dt)
        if t.name in plan:
            a = a.transpose(plan[t.name])
            if flip[t.name]:
                a = a[:, ::-1, ::-1, :]
        inits.append(ow.tensor(t.name, np.ascontiguousarray(a)))
    nodes = []
    for n in m.node:
        attrs = b"".join(ow.f_bytes(5, a.raw) for a in n.attribute)
        body = b"".join(ow.f_bytes(1, i) for i in n.input) + b"".join(ow.f_bytes(2, o) for o in n.output)
        if n.name:
            body += ow.f_bytes(3, n.name)
        nodes.append(body + ow.f_bytes(4, n.op_type) + attrs)
    init_names = {t.name for t in m.initializer}
    ins = [ow.value_info(v.name, elem, v.dims) for v in m.input if v.name not in init_names]
    outs = [ow.value_info(v.name, elem, v.dims) for v in m.output]
    g = ow.graph(nodes, m.graph_name, ins, outs, inits)
    return ow.model(g, producer_name="ONNX2MPS", producer_version="1.0.0", ir_version=m.ir_version or 5, opset=m.opset or 9)


def optimize_model(onnx_bytes: bytes, half: bool = False) -> bytes:   # ONNX2MPS.py:104-109
    m = orr.parse_model(onnx_bytes)
    m = fuse_bn_into_conv(m)
    return model_to_mps(m, half)


def main():
    ap = argparse.ArgumentParser(description="Convert ONNX model to the ONNX2MPS flavour Smelter / smelter_amd load")
    ap.add_argument("--half", action="store_true", help="Use FP16 weights")
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    a = ap.parse_args()
    with open(a.input, "rb") as f:
        data = f.read()
    with open(a.output, "wb") as f:
        f.write(optimize_model(data, a.half))
    print("Success")


if __name__ == "__main__":
    main()
