"""AddressSanitizer + UBSan run of the product's HOST sources (the reference's own latent UB sits in the same spot:
Extensions/Foundation/Array+Extensions.swift:100-110 hands MPS a pointer into a temporary).  `make asan` builds csrc/host/*.cpp with
ROCm's clang as a plain x86-64 compiler (kernel entry points stubbed) into build/asan/host_fuzz; this test feeds it every zoo model in
both flavours, the hand-made error cases of test_host_cpu.py and seeded structured mutations of the small models.  CPU only — nothing
here touches a GPU, and no GPU sanitizer exists on this pool."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from smelter_amd import onnx2mps, onnx_writer as ow, zoo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def fuzz_bin():
    if not (os.path.exists(CLANG) and shutil.which("make")):
        pytest.skip("no clang to build the sanitizer target")
    subprocess.check_call(["make", "-s", "-C", ROOT, "asan"])
    return os.path.join(ROOT, "build", "asan", "host_fuzz")


def _hand_made():
    """small valid and invalid graphs: every converter error branch that reads attributes / tensors defensively"""
    out = []

    def one(op, inits=(), n_out=1, dims=(1, 4, 6, 6), **attrs):
        b = ow.GraphBuilder("g"); b.add_input("input", list(dims))
        names = ["input"] + [b.add_init(n, a) for n, a in inits]
        y = b.add_node(op, names, n_out=n_out, **attrs)
        b.add_output(y if n_out == 1 else y[0], list(dims))
        return b.to_model()
    w = np.ones((8, 4, 3, 3), np.float32)
    conv = dict(dilations=[1, 1], group=1, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1])
    out.append(one("Conv", [("w", w)], **conv))
    out.append(one("Conv", [("w", w)], **dict(conv, kernel_shape=[5, 5])))
    out.append(one("Conv", [("w", w.astype(np.int64))], **conv))
    out.append(one("Conv", [("w", w), ("b", np.ones(3, np.float32))], **conv))
    out.append(one("Conv", [("w", w)], **dict(conv, dilations=[3, 2], pads=[0, 0, 0, 0])))
    out.append(one("Conv", [("w", w[:, :1])], **dict(conv, group=4)))
    out.append(one("ConvTranspose", [("w", np.ones((4, 8, 2, 2), np.float32))], dilations=[1, 1], group=1, kernel_shape=[2, 2], pads=[0, 0, 0, 0], strides=[2, 2]))
    out.append(one("Reshape", [("s", np.array([-1, -1, 0], np.int64))]))
    out.append(one("Reshape", [("s", np.array([0, 0, 0, 0, 0], np.int64))]))
    out.append(one("Reshape", [("s", np.array([], np.int64))]))
    out.append(one("Upsample", [("s", np.array([1, 1, 0, -3], np.float32))], mode=b"nearest"))
    out.append(one("Upsample", mode=b"cubic", scales=[1.0, 1.0, 2.0, 2.0]))
    out.append(one("MaxPool", kernel_shape=[0, 0], pads=[0, 0, 0, 0], strides=[1, 1]))
    out.append(one("MaxPool", kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[0, 2]))
    out.append(one("Pad", pads=[0, 0, 1, 1]))
    out.append(one("Pad", pads=[0, 0, -1, 1, 0, 0, 1, 1]))
    out.append(one("Dropout", n_out=2, ratio=0.5))
    out.append(one("PRelu", [("a", np.zeros(0, np.float32))]))
    out.append(one("BatchNormalization", [(n, np.ones(c, np.float32)) for n, c in (("g", 4), ("b", 4), ("m", 3), ("v", 4))]))
    out.append(one("MatMul", [("w", np.ones((144, 3), np.float32))]))
    out.append(one("Gemm", [("w", np.ones((3,), np.float32))]))
    out.append(one("Flatten", axis=2))
    out.append(one("custom_group_norm", [("n", np.array([], np.int64)), ("g", np.ones(4, np.float32)), ("b", np.ones(4, np.float32))]))
    out.append(one("Concat", axis=1))
    out.append(b"")
    out.append(b"\x0a\xff\xff\xff\xff\x0f")
    return out


def test_host_sources_under_asan_ubsan(fuzz_bin, tmp_path):
    models = {
        "toy": zoo.toy(), "toy_mps16": onnx2mps.optimize_model(zoo.toy(), half=True),
        "reshape": zoo.reshape_net(), "reshape_mps16": onnx2mps.optimize_model(zoo.reshape_net(), half=True), "reshape_last": zoo.reshape_net(final="reshape"),
        "misc": zoo.misc_ops_net(), "deconv": zoo.deconv_net(), "deconv_mps16": onnx2mps.optimize_model(zoo.deconv_net(), half=True),
        "unet": zoo.unet(hw=(32, 32), base=8, levels=2), "r50": zoo.resnet50(hw=(32, 32)), "mbv2": zoo.mobilenet_v2(hw=(32, 32)),
    }
    files = []
    for k, v in models.items():
        p = tmp_path / (k + ".onnx"); p.write_bytes(v); files.append(str(p))
    for i, v in enumerate(_hand_made()):
        p = tmp_path / ("case%02d.onnx" % i); p.write_bytes(v); files.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([fuzz_bin, "--iters", "3000", "--seed", "7"] + files, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-4000:])
    assert "no sanitizer report" in out.stdout, out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
