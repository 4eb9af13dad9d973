"""ctypes face of the CPU oracle (oracle/liborc.so) — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from smelter_amd/.
See oracle/orc.h for what is restated and why parity is "unpinned" by the reference itself.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_proto.c", "orc_core.c", "orc_ops.c", "orc_graph.c", "orc.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])
    return so


class Shape(C.Structure):
    _fields_ = [("channels", C.c_int), ("width", C.c_int), ("height", C.c_int), ("depth", C.c_int)]

    def tuple(self):
        return (self.channels, self.width, self.height, self.depth)


class Config(C.Structure):
    _fields_ = [("align_corners", C.c_int), ("dims_override", C.c_int * 4), ("act_fp16", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "C", "H", "W", "K", "kh", "kw", "sh", "sw", "dh", "dw",
                                       "pt", "pl", "pb", "pr", "groups", "Ho", "Wo")]


ERR = {0: "ok", -1: "unsupportedInput", -2: "unsupportedOutput", -3: "unknownNodeOpType", -4: "noSuchOutput",
       -5: "graphInternalError", -6: "insufficientInputs", -7: "inconsistentState", -8: "notEnoughAttributes",
       -9: "protoDecode", -10: "unsupportedDtype", -11: "flattenAxis"}


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR.get(code, "?"), code, msg))
        self.code = code
        self.kind = ERR.get(code, "?")
        self.detail = msg


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
        L.orc_last_error.restype = C.c_char_p
        L.orc_graph_create.argtypes = [vp, C.c_size_t, C.POINTER(Config), C.POINTER(vp)]
        L.orc_graph_destroy.argtypes = [vp]
        L.orc_graph_set_round_list.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int]
        L.orc_graph_is_mps_flavor.argtypes = [vp]
        L.orc_graph_output_shapes.argtypes = [vp, C.POINTER(Shape), C.c_int]
        L.orc_graph_input_chw.argtypes = [vp, ip, ip, ip]
        L.orc_graph_run.argtypes = [vp, vp, C.c_int, C.POINTER(vp), ip, ip, ip]
        L.orc_graph_value.argtypes = [vp, C.c_char_p, ip, ip, ip]
        L.orc_graph_value.restype = vp
        L.orc_graph_ref_shape.argtypes = [vp, C.c_char_p]
        L.orc_graph_ref_shape.restype = C.POINTER(Shape)
        L.orc_free.argtypes = [vp]
        L.orc_config_default.argtypes = [C.POINTER(Config)]
        L.orc_half_to_float.argtypes = [C.c_uint16]
        L.orc_half_to_float.restype = C.c_float
        L.orc_float_to_half.argtypes = [C.c_float]
        L.orc_float_to_half.restype = C.c_uint16
        L.orc_half_to_float_n.argtypes = [vp, vp, C.c_size_t]
        L.orc_float_to_half_n.argtypes = [vp, vp, C.c_size_t]
        L.orc_reformat_conv_weight.argtypes = [vp, vp] + [C.c_int] * 6
        L.orc_conv_padded_size.argtypes = [C.c_int] * 13 + [ip, ip]
        L.orc_pool_padded_size.argtypes = [C.c_int] * 8 + [ip, ip]
        L.orc_pool_window_start.argtypes = [C.c_int] * 4 + [ip]
        L.orc_float_array_count.argtypes = [C.c_int] * 4
        L.orc_float_array_count.restype = C.c_size_t
        L.orc_pack_float_array.argtypes = [vp] + [C.c_int] * 4 + [vp]
        L.orc_conv2d.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp]
        L.orc_conv_transpose2d.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp]
        L.orc_binary.argtypes = [C.c_int, vp, vp, vp, C.c_size_t]
        L.orc_neuron.argtypes = [C.c_int, vp, vp, C.c_size_t, C.c_float]
        L.orc_softmax.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_group_norm.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float]
        L.orc_pad2d.argtypes = [vp, vp] + [C.c_int] * 9
        L.orc_resize_nchw.argtypes = [vp, vp] + [C.c_int] * 6
        L.orc_gemm_fc.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
        L.orc_batchnorm.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_float]
        L.orc_relu.argtypes = [vp, vp, C.c_size_t, C.c_float]
        L.orc_add.argtypes = [vp, vp, vp, C.c_size_t]
        L.orc_maxpool.argtypes = [vp, vp] + [C.c_int] * 12
        L.orc_avgpool.argtypes = [vp, vp] + [C.c_int] * 12
        L.orc_global_avgpool.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
        L.orc_upsample.argtypes = [vp, vp] + [C.c_int] * 8
        L.orc_round_to_half.argtypes = [vp, C.c_size_t]
        L.orc_model_decode.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.orc_model_free.argtypes = [vp]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_acc64.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def set_acc64(on: bool):
    """conv accumulates in fp64 instead of fp32 (noise-floor / spot-check mode)"""
    lib().orc_set_acc64(int(on))


def num_threads() -> int:
    return lib().orc_num_threads()


# ---- L2 helpers ----------------------------------------------------------------------------------
def half_to_float(bits: np.ndarray) -> np.ndarray:
    bits = np.ascontiguousarray(bits, dtype=np.uint16)
    out = np.empty(bits.shape, np.float32)
    lib().orc_half_to_float_n(_p(bits), _p(out), bits.size)
    return out


def float_to_half(x: np.ndarray) -> np.ndarray:
    x = _f32(x)
    out = np.empty(x.shape, np.uint16)
    lib().orc_float_to_half_n(_p(x), _p(out), x.size)
    return out


def reformat_conv_weight(w: np.ndarray, oc, ic, kh, kw, is_transpose=False) -> np.ndarray:
    w = np.ascontiguousarray(w)
    assert w.dtype.itemsize in (2, 4)
    out = np.empty_like(w)
    lib().orc_reformat_conv_weight(_p(w), _p(out), w.dtype.itemsize, oc, ic, kh, kw, int(is_transpose))
    return out


def conv_padded_size(in_w, in_h, k, s, pads, out_pad=(0, 0), is_transpose=False):
    ow, oh = C.c_int(), C.c_int()
    lib().orc_conv_padded_size(in_w, in_h, k[0], k[1], s[0], s[1], pads[0], pads[1], pads[2], pads[3],
                               out_pad[0], out_pad[1], int(is_transpose), C.byref(ow), C.byref(oh))
    return ow.value, oh.value


def pool_padded_size(in_w, in_h, k, s, p):
    ow, oh = C.c_int(), C.c_int()
    lib().orc_pool_padded_size(in_w, in_h, k[0], k[1], s[0], s[1], p[0], p[1], C.byref(ow), C.byref(oh))
    return ow.value, oh.value


def pool_window_start(n_in, n_out, k, s):
    st = C.c_int()
    lib().orc_pool_window_start(n_in, n_out, k, s, C.byref(st))
    return st.value


def pack_float_array(x_nchw: np.ndarray) -> np.ndarray:
    x = _f32(x_nchw)
    n, c, h, w = x.shape
    out = np.empty(lib().orc_float_array_count(n, c, h, w), np.float32)
    lib().orc_pack_float_array(_p(x), n, c, h, w, _p(out))
    return out


# ---- per-op ---------------------------------------------------------------------------------------
def conv2d(x, w_ohwi, bias=None, stride=(1, 1), pads=(0, 0, 0, 0), dil=(1, 1), groups=1):
    x, w = _f32(x), _f32(w_ohwi)
    n, c, h, wd = x.shape
    k, kh, kw, cg = w.shape
    assert cg * groups == c
    wo, ho = conv_padded_size(wd, h, (kh, kw), stride, pads)
    d = ConvDesc(n, c, h, wd, k, kh, kw, stride[0], stride[1], dil[0], dil[1], pads[0], pads[1], pads[2], pads[3],
                 groups, ho, wo)
    y = np.empty((n, k, ho, wo), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_conv2d(C.byref(d), _p(x), _p(w), _p(b), _p(y))
    return y


def conv_transpose2d(x, w_ohwi_flipped, bias=None, stride=(1, 1), pads=(0, 0, 0, 0), dil=(1, 1), out_pad=(0, 0)):
    """MPSCNNConvolutionTransposeNode over the data source's flipped OHWI weights (reformat_conv_weight(..., is_transpose=True))"""
    x, w = _f32(x), _f32(w_ohwi_flipped)
    n, c, h, wd = x.shape
    k, kh, kw, cg = w.shape
    assert cg == c
    wo, ho = conv_padded_size(wd, h, (kh, kw), stride, pads, out_pad, is_transpose=True)
    d = ConvDesc(n, c, h, wd, k, kh, kw, stride[0], stride[1], dil[0], dil[1], pads[0], pads[1], pads[2], pads[3], 1, ho, wo)
    y = np.empty((n, k, ho, wo), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_conv_transpose2d(C.byref(d), _p(x), _p(w), _p(b), _p(y))
    return y


def gemm_fc(x, w_oi, bias=None):
    x, w = _f32(x), _f32(w_oi)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_gemm_fc(_p(x), _p(w), _p(b), _p(y), x.shape[0], w.shape[1], w.shape[0])
    return y


def batchnorm(x, gamma, beta, mean, var, eps=1e-5):
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty_like(x)
    g, b, m, v = map(_f32, (gamma, beta, mean, var))
    lib().orc_batchnorm(_p(x), _p(y), n, c, h * w, _p(g), _p(b), _p(m), _p(v), eps)
    return y


def relu(x, slope=0.0):
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_relu(_p(x), _p(y), x.size, slope)
    return y


def add(a, b):
    a, b = _f32(a), _f32(b)
    y = np.empty_like(a)
    lib().orc_add(_p(a), _p(b), _p(y), a.size)
    return y


NEURON = {"sigmoid": 3, "hardsigmoid": 4, "tanh": 5, "elu": 6, "abs": 7, "softplus": 8, "softsign": 9, "exp": 10, "log": 11, "clip": 12}
BINARY = {"add": 0, "sub": 1, "mul": 2, "div": 3}


def neuron(kind: str, x, a: float = 0.0):
    x = _f32(x); y = np.empty_like(x)
    lib().orc_neuron(NEURON[kind], _p(x), _p(y), x.size, a)
    return y


def binary(op: str, a, b):
    a, b = _f32(a), _f32(b); y = np.empty_like(a)
    lib().orc_binary(BINARY[op], _p(a), _p(b), _p(y), a.size)
    return y


def softmax(x, log: bool = False):
    x = _f32(x); n, c, h, w = x.shape; y = np.empty_like(x)
    lib().orc_softmax(_p(x), _p(y), n, c, h * w, 1 if log else 0)
    return y


def group_norm(x, groups, gamma, beta, eps=1e-5):
    x = _f32(x); n, c, h, w = x.shape; y = np.empty_like(x)
    lib().orc_group_norm(_p(x), _p(y), n, c, h * w, groups, _p(_f32(gamma)), _p(_f32(beta)), eps)
    return y


def pad2d(x, pads, mode: str = "constant"):
    x = _f32(x); n, c, h, w = x.shape; pt, pl, pb, pr = pads
    y = np.empty((n, c, h + pt + pb, w + pl + pr), np.float32)
    lib().orc_pad2d(_p(x), _p(y), n, c, h, w, pt, pl, pb, pr, {"constant": 0, "reflect": 1, "edge": 2}[mode])
    return y


def resize(x, size, mode: str = "bilinear"):
    """Configuration.inputConstraint .forceInputScale: resample NCHW planes to size = (H, W); mode 'bilinear' | 'lanczos'"""
    x = _f32(x); n, c, h, w = x.shape
    y = np.empty((n, c, size[0], size[1]), np.float32)
    lib().orc_resize_nchw(_p(x), _p(y), n * c, h, w, size[0], size[1], {"lanczos": 1, "bilinear": 2}[mode])
    return y


def _pool(fn, x, k, s, p):
    x = _f32(x)
    n, c, h, w = x.shape
    wo, ho = pool_padded_size(w, h, k, s, p)
    y = np.empty((n, c, ho, wo), np.float32)
    fn(_p(x), _p(y), n, c, h, w, k[0], k[1], s[0], s[1], p[0], p[1], ho, wo)
    return y


def maxpool(x, k, s, p):
    return _pool(lib().orc_maxpool, x, k, s, p)


def avgpool(x, k, s, p):
    return _pool(lib().orc_avgpool, x, k, s, p)


def global_avgpool(x):
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, 1, 1), np.float32)
    lib().orc_global_avgpool(_p(x), _p(y), n, c, h * w)
    return y


def upsample(x, sy, sx, bilinear=False, align_corners=True):
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, h * sy, w * sx), np.float32)
    lib().orc_upsample(_p(x), _p(y), n, c, h, w, sy, sx, int(bilinear), int(align_corners))
    return y


def round_to_half(x):
    x = _f32(x).copy()
    lib().orc_round_to_half(_p(x), x.size)
    return x


# ---- graph ------------------------------------------------------------------------------------------
class Graph:
    """Oracle counterpart of ONNXGraph(data:configuration:) + encode + toFloatArray."""

    def __init__(self, onnx_bytes: bytes, align_corners: bool = True, dims: dict | None = None, act_fp16: bool = False,
                 input_constraint: str = "none"):
        L = lib()
        self.input_constraint = input_constraint      # Configuration.inputConstraint: "none" | "lanczos" | "bilinear"
        cfg = Config()
        L.orc_config_default(C.byref(cfg))
        cfg.align_corners = int(align_corners)
        cfg.act_fp16 = int(act_fp16)
        for k, v in (dims or {}).items():
            cfg.dims_override[k] = v
        self._h = C.c_void_p()
        self._buf = bytes(onnx_bytes)
        rc = L.orc_graph_create(self._buf, len(self._buf), C.byref(cfg), C.byref(self._h))
        if rc:
            raise OracleError(rc, L.orc_last_error().decode())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _LIB is not None:
            _LIB.orc_graph_destroy(h)

    def set_round_list(self, names):
        """emulate fp16 storage for exactly these value names (an executor's materialised tensors)"""
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        lib().orc_graph_set_round_list(self._h, arr, len(names))

    @property
    def is_mps_flavor(self) -> bool:
        return bool(lib().orc_graph_is_mps_flavor(self._h))

    @property
    def output_shapes(self):
        arr = (Shape * 8)()
        n = lib().orc_graph_output_shapes(self._h, arr, 8)
        return [arr[i].tuple() for i in range(min(n, 8))]

    @property
    def input_chw(self):
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        rc = lib().orc_graph_input_chw(self._h, C.byref(c), C.byref(h), C.byref(w))
        if rc:
            raise OracleError(rc, lib().orc_last_error().decode())
        return c.value, h.value, w.value

    def run(self, x_nchw: np.ndarray) -> np.ndarray:
        x = _f32(x_nchw)
        _, ih, iw = self.input_chw
        if x.shape[2:] != (ih, iw):                   # .forceInputScale puts a scale node in front (ONNXGraph.swift:219-241)
            if self.input_constraint == "none":
                raise OracleError(-1, "source image %dx%d, model takes %dx%d, inputConstraint .none" % (x.shape[2], x.shape[3], ih, iw))
            x = resize(x, (ih, iw), self.input_constraint)
        L = lib()
        out = C.c_void_p()
        oc, oh, ow = C.c_int(), C.c_int(), C.c_int()
        rc = L.orc_graph_run(self._h, _p(x), x.shape[0], C.byref(out), C.byref(oc), C.byref(oh), C.byref(ow))
        if rc:
            raise OracleError(rc, L.orc_last_error().decode())
        shape = (x.shape[0], oc.value, oh.value, ow.value)
        n = int(np.prod(shape))
        res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_float)), (n,)).reshape(shape).copy()
        L.orc_free(out)
        return res

    def value(self, name: str):
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        p = lib().orc_graph_value(self._h, name.encode(), C.byref(c), C.byref(h), C.byref(w))
        if not p:
            return None
        return (p, c.value, h.value, w.value)

    def ref_shape(self, name: str):
        p = lib().orc_graph_ref_shape(self._h, name.encode())
        return p.contents.tuple() if p else None
