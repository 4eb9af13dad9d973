"""Kernel-level operators: thin torch-tensor wrappers over the `sm_*` C ABI (include/smelter_hip.h).

torch supplies device memory and the stream only; every FLOP runs in the hand-written gfx950 kernels.
Activations are NHWC fp16 `torch.Tensor`s on the GPU ([N,H,W,C], contiguous or with a channel-slice view
whose pixel stride is passed as `ld`); conv weights are OHWI fp16; bias fp32.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._native import ConvDesc, PoolDesc, StagePhase, check, lib

ACT_NONE, ACT_RELU, ACT_PRELU, ACT_CLIP = 0, 1, 2, 12


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _ld(t: torch.Tensor) -> int:
    """pixel stride (elements) of an NHWC tensor or channel-slice view"""
    assert t.stride(-1) == 1 or t.shape[-1] == 1, "channel axis must be innermost"
    n, h, w, c = t.shape
    if w > 1:                                      # size-1 axes carry arbitrary strides in torch/numpy
        return t.stride(2)
    if h > 1:
        return t.stride(1)
    if n > 1:
        return t.stride(0)
    return c


def _check_nhwc(t: torch.Tensor):
    assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 4
    n, h, w, c = t.shape
    ld = _ld(t)
    assert (h == 1 or t.stride(1) == w * ld) and (n == 1 or t.stride(0) == h * w * ld), "only the channel axis may be sliced"
    return n, h, w, c, ld


def conv_out_size(h, w, kh, kw, sh, sw, pads):
    ow, oh = C.c_int32(), C.c_int32()
    lib().sm_conv_padded_size(w, h, kh, kw, sh, sw, pads[0], pads[1], pads[2], pads[3], 0, 0, 0, C.byref(ow), C.byref(oh))
    return oh.value, ow.value


CFG_LOG = None      # tests/conftest.py sets this to a set(): every tile configuration a *_cfg entry point has run (the call returned SM_OK)


def _ran(cfg):
    if CFG_LOG is not None and cfg is not None and int(cfg) >= 0:
        CFG_LOG.add(int(cfg))


def conv2d(x, w_ohwi, bias=None, residual=None, stride=(1, 1), pads=(0, 0, 0, 0), dilation=(1, 1), groups=1,
           act=ACT_NONE, slope=0.0, out=None, cfg=None, workspace=None, ups=False):
    """ups=True: `x` is the half-resolution tensor, read through a fused nearest 2x upsampling (sm_conv_desc.x_ups)"""
    n, h, w, cin, x_ld = _check_nhwc(x)
    if ups:
        h, w = 2 * h, 2 * w
    cout, kh, kw, cg = w_ohwi.shape
    assert w_ohwi.dtype == torch.float16 and w_ohwi.is_contiguous() and cg * groups == cin
    ho, wo = conv_out_size(h, w, kh, kw, stride[0], stride[1], pads)
    if out is None:
        out = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x.device)
    _, oh_, ow_, oc_, y_ld = _check_nhwc(out)
    assert (oh_, ow_, oc_) == (ho, wo, cout)
    res_ld = 0
    if residual is not None:
        _, rh, rw, rc, res_ld = _check_nhwc(residual)
        assert (rh, rw, rc) == (ho, wo, cout)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == cout
    d = ConvDesc(n, h, w, cin, cout, kh, kw, stride[0], stride[1], dilation[0], dilation[1], pads[0], pads[1], pads[2],
                 pads[3], groups, ho, wo, act, slope, x_ld, y_ld, res_ld)
    d.x_ups = 1 if ups else 0
    L = lib()
    if cfg is not None:
        wsb = workspace.numel() * workspace.element_size() if workspace is not None else 0
        check(L.sm_conv2d_fwd_cfg(C.byref(d), cfg, _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(residual), _ptr(out), _ptr(workspace), wsb,
                                  _stream()))
        _ran(cfg)
    elif workspace is not None:
        check(L.sm_conv2d_fwd_ws(C.byref(d), _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(residual), _ptr(out), _ptr(workspace),
                                 workspace.numel() * workspace.element_size(), _stream()))
    else:
        check(L.sm_conv2d_fwd(C.byref(d), _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(residual), _ptr(out), _stream()))
    return out


def conv2d_dual(x, x2, w_ohwi, bias=None, stride2=1, act=ACT_NONE, slope=0.0, cfg=-1, out=None):
    """pointwise convolution over two sources concatenated along K (sm_conv2d_dual_fwd_cfg): y = act(w[:, :K1] . x + w[:, K1:] . x2[:, ::stride2, ::stride2] + bias)"""
    n, h, w, k1, x_ld = _check_nhwc(x)
    n2, h2, w2, k2, x2_ld = _check_nhwc(x2)
    cout, kh, kw, kk = w_ohwi.shape
    assert n2 == n and (kh, kw) == (1, 1) and kk == k1 + k2 and w_ohwi.dtype == torch.float16 and w_ohwi.is_contiguous()
    if out is None:
        out = torch.empty((n, h, w, cout), dtype=torch.float16, device=x.device)
    d = ConvDesc(n, h, w, k1 + k2, cout, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, h, w, act, slope, x_ld, _ld(out), 0)
    check(lib().sm_conv2d_dual_fwd_cfg(C.byref(d), cfg, _ptr(x), _ptr(x2), k2, x2_ld, h2, w2, stride2, _ptr(w_ohwi), _ptr(bias), _ptr(out), _stream()))
    _ran(cfg)
    return out


class Stage:
    """A run of convolutions as ONE persistent launch (sm_stage_fwd).  `add(...)` appends a phase with the operands of `conv2d` / `conv2d_dual`;
    tensors are kept alive by the object.  `run()` enqueues the launch, `status()` reads (error_bits, mode) after synchronising."""

    def __init__(self, four_waves=False):
        """four_waves: keep the planner to the 4-wave tile forms (default: the 8-wave forms when every phase has one)"""
        self.phases, self.keep, self.ctl, self.four_waves = [], [], None, four_waves

    def add(self, x, w_ohwi, bias=None, residual=None, act=ACT_NONE, slope=0.0, out=None, x2=None, stride2=1, kind=None):
        n, h, w, k1, x_ld = _check_nhwc(x)
        cout, kh, kw, kk = w_ohwi.shape
        k2 = 0
        if x2 is not None:
            n2, h2, w2, k2, x2_ld = _check_nhwc(x2)
            assert n2 == n
        assert kk == k1 + k2 and w_ohwi.dtype == torch.float16 and w_ohwi.is_contiguous()
        pad = (kh - 1) // 2
        if out is None:
            out = torch.empty((n, h, w, cout), dtype=torch.float16, device=x.device)
        res_ld = _ld(residual) if residual is not None else 0
        ph = StagePhase()
        ph.d = ConvDesc(n, h, w, k1 + k2, cout, kh, kw, 1, 1, 1, 1, pad, pad, pad, pad, 1, h, w, act, slope, x_ld, _ld(out), res_ld)
        ph.x, ph.w_ohwi, ph.bias, ph.residual, ph.y = x.data_ptr(), w_ohwi.data_ptr(), bias.data_ptr() if bias is not None else None, \
            residual.data_ptr() if residual is not None else None, out.data_ptr()
        if x2 is not None:
            ph.x2, ph.K2, ph.x2_ld, ph.x2_H, ph.x2_W, ph.x2_stride = x2.data_ptr(), k2, x2_ld, h2, w2, stride2
        ph.form = (-1 if self.four_waves else 0) if kind is None else kind + 1
        self.phases.append(ph)
        self.keep += [x, w_ohwi, bias, residual, out, x2]
        return out

    def _array(self, sel=None):
        ph = self.phases if sel is None else [self.phases[i] for i in sel]
        return (StagePhase * len(ph))(*ph), len(ph)

    def supported(self):
        arr, n = self._array()
        return bool(lib().sm_stage_supported(arr, n))

    def kinds(self):
        arr, n = self._array()
        return [lib().sm_stage_phase_kind(arr, n, i) for i in range(n)]

    def pack(self):
        """re-order every phase's weights for the tile form the planner gave it (sm_stage_pack_weights, host side) and hand the kernel the packed copies"""
        arr, n = self._array()
        self.packed = []
        ws = [t for t in self.keep[1::6]]
        for i in range(n):
            nb = lib().sm_stage_packed_weight_bytes(arr, n, i)
            wh = ws[i].cpu().contiguous()
            dst = torch.empty(nb // 2, dtype=torch.float16)
            check(lib().sm_stage_pack_weights(arr, n, i, C.c_void_p(wh.data_ptr()), C.c_void_p(dst.data_ptr())))
            d = dst.cuda()
            self.packed.append(d)
            self.phases[i].w_packed = d.data_ptr()

    def run(self, sel=None):
        """sel: indices of the phases to launch as one run (default: all)"""
        arr, n = self._array(sel)
        if self.ctl is None:
            full, nf = self._array()
            self.ctl = torch.zeros(lib().sm_stage_ctl_bytes(full, nf) // 4, dtype=torch.int32, device="cuda")
        check(lib().sm_stage_fwd(arr, n, _ptr(self.ctl), _stream()))

    def status(self):
        e, m = C.c_int32(), C.c_int32()
        check(lib().sm_stage_status(_ptr(self.ctl), _stream(), C.byref(e), C.byref(m)))
        return e.value, m.value


def stem_conv(x_nchw, w_ohwi_f32, bias=None, stride=1, pads=(0, 0, 0, 0), act=ACT_NONE, slope=0.0, cfg=None):
    """First-layer convolution of a C <= 4 fp32 NCHW image the way the executor runs it (executor.cpp "image stem"): row-pair image
    (sm_nchw_f32_to_rowpair_f16) + weights repacked to [Cout][ceil(kH/2)][8 pixels][2 rows][4 channels], launched as a Cin = 64, kW = 1,
    row-dilation-2 convolution with x_ld = 8.  Returns NHWC fp16."""
    assert x_nchw.is_cuda and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    n, c, h, w = x_nchw.shape
    cout, kh, kw, c2 = w_ohwi_f32.shape
    assert c == c2 and c <= 4 and kw <= 8
    s, P = stride, (kh + 1) // 2
    ho, wo = conv_out_size(h, w, kh, kw, s, s, pads)
    hs, ws = (ho - 1) * s + 2 * (P - 1) + 1, (wo - 1) * s + 8
    f32_path = cfg is not None and lib().sm_conv2d_config_loader_waves(cfg) > 0   # loader-wave configurations: the kernel builds its patches from the fp32 batch itself
    if not f32_path:
        img = torch.empty((n, hs, ws, 8), dtype=torch.float16, device=x_nchw.device)
        check(lib().sm_nchw_f32_to_rowpair_f16(_ptr(x_nchw), _ptr(img), n, c, h, w, hs, ws, pads[0], pads[1], _stream()))
    wp = torch.zeros((cout, P, 8, 2, 4), dtype=torch.float32)
    wc = w_ohwi_f32.detach().cpu().float()
    for r in range(kh):
        wp[:, r // 2, :kw, r & 1, :c] = wc[:, r]
    wp = wp.reshape(cout, P, 1, 64).to(torch.float16).to(x_nchw.device).contiguous()
    out = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x_nchw.device)
    d = ConvDesc(n, hs, ws, 64, cout, P, 1, s, s, 2, 1, 0, 0, 0, 0, 1, ho, wo, act, slope, 8, cout, 0)
    if f32_path:
        check(lib().sm_conv2d_stem_f32_fwd_cfg(C.byref(d), cfg, _ptr(x_nchw), c, h, w, pads[0], pads[1], _ptr(wp), _ptr(bias), _ptr(out), _stream()))
        _ran(cfg)
    elif cfg is None:
        check(lib().sm_conv2d_fwd(C.byref(d), _ptr(img), _ptr(wp), _ptr(bias), None, _ptr(out), _stream()))
    else:
        check(lib().sm_conv2d_fwd_cfg(C.byref(d), cfg, _ptr(img), _ptr(wp), _ptr(bias), None, _ptr(out), None, 0, _stream()))
        _ran(cfg)
    return out


def stem_conv_pool(x_nchw, w_ohwi_f32, bias=None, stride=2, pads=(3, 3, 3, 3), act=ACT_RELU, slope=0.0, pool_pad=1, out=None):
    """Image stem + MaxPool 3x3 / stride 2 (padding pool_pad, the reference's window-start rule) in one launch (sm_conv2d_stem_f32_pool_fwd):
    returns the pooled NHWC fp16 tensor; the stem's own output is never stored."""
    assert x_nchw.is_cuda and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    n, c, h, w = x_nchw.shape
    cout, kh, kw, c2 = w_ohwi_f32.shape
    assert c == c2 and c <= 4 and kw <= 8
    s, P = stride, (kh + 1) // 2
    ho, wo = conv_out_size(h, w, kh, kw, s, s, pads)
    hs, ws = (ho - 1) * s + 2 * (P - 1) + 1, (wo - 1) * s + 8
    L = lib()
    pw_, ph_ = C.c_int32(), C.c_int32()
    L.sm_pool_padded_size(wo, ho, 3, 3, 2, 2, pool_pad, pool_pad, C.byref(pw_), C.byref(ph_))
    assert L.sm_pool_window_start(ho, ph_.value, 3, 2) == 0 and L.sm_pool_window_start(wo, pw_.value, 3, 2) == 0, "fused pooling: windows start at 2p"
    wp = torch.zeros((cout, P, 8, 2, 4), dtype=torch.float32)
    wc = w_ohwi_f32.detach().cpu().float()
    for r in range(kh):
        wp[:, r // 2, :kw, r & 1, :c] = wc[:, r]
    wp = wp.reshape(cout, P, 1, 64).to(torch.float16).to(x_nchw.device).contiguous()
    if out is None:
        out = torch.empty((n, ph_.value, pw_.value, cout), dtype=torch.float16, device=x_nchw.device)
    d = ConvDesc(n, hs, ws, 64, cout, P, 1, s, s, 2, 1, 0, 0, 0, 0, 1, ho, wo, act, slope, 8, cout, 0)
    check(L.sm_conv2d_stem_f32_pool_fwd(C.byref(d), _ptr(x_nchw), c, h, w, pads[0], pads[1], _ptr(wp), _ptr(bias), _ptr(out), ph_.value, pw_.value, _ld(out), _stream()))
    return out


def conv2d_pool(x, w_ohwi, bias, act=ACT_RELU, slope=0.0, cfg=-1):
    """3x3 / s1 / p1 convolution + 2x2 / s2 max-pool of its output in one launch (sm_conv2d_pool_fwd_cfg); returns (y, pooled)"""
    n, h, w, cin, x_ld = _check_nhwc(x)
    cout = w_ohwi.shape[0]
    y = torch.empty((n, h, w, cout), dtype=torch.float16, device=x.device)
    yp = torch.empty((n, h // 2, w // 2, cout), dtype=torch.float16, device=x.device)
    d = ConvDesc(n, h, w, cin, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, h, w, act, slope, x_ld, _ld(y), 0)
    check(lib().sm_conv2d_pool_fwd_cfg(C.byref(d), cfg, _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(y), _ptr(yp), _ld(yp), _stream()))
    _ran(cfg)
    return y, yp


def conv2d_head(x, w_ohwi, bias, w3, b3, act=ACT_RELU, slope=0.0, act3=ACT_NONE, slope3=0.0, cfg=-1, store_y=True):
    """3x3 / s1 / p1 convolution + the 1x1 convolution that consumes it (n3 <= 32 channels) in one launch (sm_conv2d_head_fwd_cfg);
    returns (y or None, y3)"""
    n, h, w, cin, x_ld = _check_nhwc(x)
    cout, n3 = w_ohwi.shape[0], w3.shape[0]
    y = torch.empty((n, h, w, cout), dtype=torch.float16, device=x.device) if store_y else None
    y3 = torch.empty((n, h, w, n3), dtype=torch.float16, device=x.device)
    d = ConvDesc(n, h, w, cin, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, h, w, act, slope, x_ld, cout, 0)
    check(lib().sm_conv2d_head_fwd_cfg(C.byref(d), cfg, _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(y), _ptr(w3), _ptr(b3), n3, act3, slope3, _ptr(y3), _ld(y3), _stream()))
    _ran(cfg)
    return y, y3


def conv2d_xhead(x, w_ohwi, bias, w3, b3, residual=None, act=ACT_RELU, slope=0.0, act3=ACT_RELU, slope3=0.0, cfg=-1, store_y=False):
    """3x3 / s1 / p1 convolution (64 -> 64) + the 1x1 expand convolution behind it (n3 <= 256) + residual + activation in one launch
    (sm_conv2d_xhead_fwd_cfg); returns (y or None, y3)"""
    n, h, w, cin, x_ld = _check_nhwc(x)
    cout, n3 = w_ohwi.shape[0], w3.shape[0]
    y = torch.empty((n, h, w, cout), dtype=torch.float16, device=x.device) if store_y else None
    y3 = torch.empty((n, h, w, n3), dtype=torch.float16, device=x.device)
    res_ld = _check_nhwc(residual)[4] if residual is not None else 0
    d = ConvDesc(n, h, w, cin, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, h, w, act, slope, x_ld, cout, 0)
    check(lib().sm_conv2d_xhead_fwd_cfg(C.byref(d), cfg, _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(y), _ptr(w3), _ptr(b3), n3, act3, slope3,
                                        _ptr(residual), res_ld, _ptr(y3), _ld(y3), _stream()))
    _ran(cfg)
    return y, y3


def conv2d_xhead_proj(x, w_ohwi, bias, w3, b3sum, xs, ws, act=ACT_RELU, slope=0.0, act3=ACT_RELU, slope3=0.0, cfg=-1):
    """3x3 / s1 / p1 convolution (64 -> 64) + 1x1 expand (w3) + 1x1 projection of the 64-channel block input xs (ws) + Add + activation in one
    launch (sm_conv2d_xhead_proj_fwd_cfg); b3sum = the two 1x1 biases added; returns y3"""
    n, h, w, cin, x_ld = _check_nhwc(x)
    cout, n3 = w_ohwi.shape[0], w3.shape[0]
    y3 = torch.empty((n, h, w, n3), dtype=torch.float16, device=x.device)
    d = ConvDesc(n, h, w, cin, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, h, w, act, slope, x_ld, cout, 0)
    check(lib().sm_conv2d_xhead_proj_fwd_cfg(C.byref(d), cfg, _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(w3), _ptr(b3sum), n3, act3, slope3,
                                             _ptr(xs), _check_nhwc(xs)[4], _ptr(ws), _ptr(y3), _ld(y3), _stream()))
    _ran(cfg)
    return y3


def bottleneck(x, w1, b1, w2_ohwi, b2, w3, b3, residual=None, acts=(ACT_RELU, ACT_RELU, ACT_RELU), slopes=(0.0, 0.0, 0.0), out=None):
    """a whole bottleneck block in one launch (sm_bottleneck_fwd): 1x1 reduce w1 [64][C0] -> 3x3 / s1 / p1 w2 [64][3][3][64] -> 1x1 expand w3 [C3][64] (+ residual), the two
    hidden tensors in LDS only"""
    from ._native import BneckDesc
    n, h, w, c0, x_ld = _check_nhwc(x)
    cm, c3 = w1.shape[0], w3.shape[0]
    y = out if out is not None else torch.empty((n, h, w, c3), dtype=torch.float16, device=x.device)
    res_ld = _check_nhwc(residual)[4] if residual is not None else 0
    d = BneckDesc(n, h, w, c0, cm, c3, x_ld, _ld(y), res_ld, acts[0], slopes[0], acts[1], slopes[1], acts[2], slopes[2])
    check(lib().sm_bottleneck_fwd(C.byref(d), _ptr(x), _ptr(w1), _ptr(b1), _ptr(w2_ohwi), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(residual), _ptr(y), _stream()))
    return y


def mbconv(x, w1, b1, wd, bd, w2, b2, stride=1, residual=False, acts=(ACT_RELU, ACT_RELU, ACT_NONE), slopes=(0.0, 0.0, 0.0), kernel=0):
    """fused inverted-residual block (sm_mbconv_fwd): [1x1 expand w1 (None = absent)] -> depthwise 3x3 wd [Ch,3,3] -> 1x1 project w2 (+ x)"""
    from ._native import MbConvDesc
    n, h, w, cin, x_ld = _check_nhwc(x)
    ch, cout = wd.shape[0], w2.shape[0]
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x.device)
    d = MbConvDesc(n, h, w, ho, wo, stride, cin, ch, cout, 1 if w1 is not None else 0, 1 if residual else 0, acts[0], slopes[0], acts[1], slopes[1], acts[2], slopes[2], x_ld, _ld(y))
    # kernel = 1 .. 4: that block kernel where its domain holds the block (sm_mbconv_fwd_kind), 0 = the launcher's choice
    check(lib().sm_mbconv_fwd_kind(C.byref(d), kernel, _ptr(x), _ptr(w1), _ptr(b1), _ptr(wd), _ptr(bd), _ptr(w2), _ptr(b2), _ptr(y), _stream()))
    return y


def conv_transpose2d(x, w_ohwi_flipped, bias=None, stride=(1, 1), pads=(0, 0, 0, 0), dilation=(1, 1), out_pad=(0, 0), act=ACT_NONE, slope=0.0):
    """sm_conv_transpose2d_fwd: direct kernel over the flipped OHWI weights the reference's ConvDataSource produces"""
    n, h, w, cin, x_ld = _check_nhwc(x)
    cout, kh, kw, cg = w_ohwi_flipped.shape
    assert w_ohwi_flipped.dtype == torch.float16 and w_ohwi_flipped.is_contiguous() and cg == cin
    ho = (h - 1) * stride[0] - pads[0] - pads[2] + kh + out_pad[0]
    wo = (w - 1) * stride[1] - pads[1] - pads[3] + kw + out_pad[1]
    out = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x.device)
    d = ConvDesc(n, h, w, cin, cout, kh, kw, stride[0], stride[1], dilation[0], dilation[1], pads[0], pads[1], pads[2], pads[3], 1, ho, wo,
                 act, slope, x_ld, cout, 0)
    check(lib().sm_conv_transpose2d_fwd(C.byref(d), _ptr(x), _ptr(w_ohwi_flipped), _ptr(bias), _ptr(out), _stream()))
    return out


def conv2d_time(x, w_ohwi, bias=None, residual=None, stride=(1, 1), pads=(0, 0, 0, 0), groups=1, act=ACT_NONE, cfg=-1, iters=20, out=None, workspace=None):
    """average ms per launch of one conv shape / tile config, timed inside the library (no Python launch overhead)"""
    n, h, w, cin, x_ld = _check_nhwc(x)
    cout, kh, kw, cg = w_ohwi.shape
    ho, wo = conv_out_size(h, w, kh, kw, stride[0], stride[1], pads)
    if out is None:
        out = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x.device)
    d = ConvDesc(n, h, w, cin, cout, kh, kw, stride[0], stride[1], 1, 1, pads[0], pads[1], pads[2], pads[3], groups, ho, wo, act, 0.0,
                 x_ld, _ld(out), _ld(residual) if residual is not None else 0)
    ms = C.c_float()
    wsb = workspace.numel() * workspace.element_size() if workspace is not None else 0
    check(lib().sm_conv2d_time(C.byref(d), cfg, _ptr(x), _ptr(w_ohwi), _ptr(bias), _ptr(residual), _ptr(out), _ptr(workspace), wsb, iters, _stream(), C.byref(ms)))
    return ms.value


def conv2d_workspace_bytes(x_shape, w_shape, stride=(1, 1), pads=(0, 0, 0, 0), groups=1) -> int:
    n, h, w, cin = x_shape
    cout, kh, kw, _ = w_shape
    ho, wo = conv_out_size(h, w, kh, kw, stride[0], stride[1], pads)
    d = ConvDesc(n, h, w, cin, cout, kh, kw, stride[0], stride[1], 1, 1, pads[0], pads[1], pads[2], pads[3], groups, ho, wo,
                 0, 0.0, 0, 0, 0)
    return lib().sm_conv2d_workspace_bytes(C.byref(d))


def config_name(cfg: int) -> str:
    """label of a tile configuration (sm_conv2d_config_name)"""
    return lib().sm_conv2d_config_name(int(cfg)).decode()


def num_conv_configs() -> int:
    return lib().sm_conv2d_num_configs()


def batchnorm(x, scale, shift, act=ACT_NONE, slope=0.0, out=None):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=x.device) if out is None else out
    check(lib().sm_batchnorm(_ptr(x), _ptr(out), n * h * w, c, x_ld, _ld(out), _ptr(scale), _ptr(shift), act, slope, _stream()))
    return out


def add(a, b, act=ACT_NONE, slope=0.0, out=None):
    n, h, w, c, a_ld = _check_nhwc(a)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=a.device) if out is None else out
    check(lib().sm_add(_ptr(a), _ptr(b), _ptr(out), n * h * w, c, a_ld, _ld(b), _ld(out), act, slope, _stream()))
    return out


ACT_CODES = {"none": 0, "relu": 1, "prelu": 2, "sigmoid": 3, "hardsigmoid": 4, "tanh": 5, "elu": 6, "abs": 7, "softplus": 8, "softsign": 9, "exp": 10, "log": 11, "clip": 12}
BIN_CODES = {"add": 0, "sub": 1, "mul": 2, "div": 3}
PAD_CODES = {"constant": 0, "reflect": 1, "edge": 2}


def binary(op, a, b, act=ACT_NONE, slope=0.0):
    n, h, w, c, a_ld = _check_nhwc(a)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=a.device)
    check(lib().sm_binary(BIN_CODES[op], _ptr(a), _ptr(b), _ptr(out), n * h * w, c, a_ld, _ld(b), _ld(out), act, slope, _stream()))
    return out


def neuron(kind, x, a=0.0):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=x.device)
    check(lib().sm_neuron(_ptr(x), _ptr(out), n * h * w, c, x_ld, _ld(out), ACT_CODES[kind], a, _stream()))
    return out


def softmax(x, log=False):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=x.device)
    check(lib().sm_softmax(_ptr(x), _ptr(out), n * h * w, c, x_ld, _ld(out), 1 if log else 0, _stream()))
    return out


def group_norm(x, groups, gamma, beta, eps=1e-5):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=x.device)
    check(lib().sm_group_norm(_ptr(x), _ptr(out), n, h * w, c, groups, _ptr(gamma), _ptr(beta), eps, x_ld, _ld(out), _stream()))
    return out


def pad2d(x, pads, mode="constant"):
    n, h, w, c, x_ld = _check_nhwc(x)
    pt, pl, pb, pr = pads
    out = torch.empty((n, h + pt + pb, w + pl + pr, c), dtype=torch.float16, device=x.device)
    check(lib().sm_pad2d(_ptr(x), _ptr(out), n, h, w, c, pt, pl, pb, pr, PAD_CODES[mode], x_ld, _ld(out), _stream()))
    return out


def relu(x, slope=0.0, out=None):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=x.device) if out is None else out
    check(lib().sm_relu(_ptr(x), _ptr(out), n * h * w, c, x_ld, _ld(out), slope, _stream()))
    return out


def _pool(fn, x, k, s, p, out):
    n, h, w, c, x_ld = _check_nhwc(x)
    L = lib()
    ow, oh = C.c_int32(), C.c_int32()
    L.sm_pool_padded_size(w, h, k[0], k[1], s[0], s[1], p[0], p[1], C.byref(ow), C.byref(oh))
    ho, wo = oh.value, ow.value
    out = torch.empty((n, ho, wo, c), dtype=torch.float16, device=x.device) if out is None else out
    d = PoolDesc(n, h, w, c, k[0], k[1], s[0], s[1], L.sm_pool_window_start(h, ho, k[0], s[0]),
                 L.sm_pool_window_start(w, wo, k[1], s[1]), ho, wo, x_ld, _ld(out))
    check(fn(C.byref(d), _ptr(x), _ptr(out), _stream()))
    return out


def maxpool2d(x, k, s, p, out=None):
    return _pool(lib().sm_maxpool2d, x, k, s, p, out)


def avgpool2d(x, k, s, p, out=None):
    return _pool(lib().sm_avgpool2d, x, k, s, p, out)


def global_avgpool(x, out=None):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, 1, 1, c), dtype=torch.float16, device=x.device) if out is None else out
    check(lib().sm_global_avgpool(_ptr(x), _ptr(out), n, h * w, c, x_ld, _ld(out), _stream()))
    return out


def upsample(x, sy, sx, bilinear=False, align_corners=True, out=None):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, h * sy, w * sx, c), dtype=torch.float16, device=x.device) if out is None else out
    if bilinear:
        check(lib().sm_upsample_bilinear(_ptr(x), _ptr(out), n, h, w, c, sy, sx, int(align_corners), x_ld, _ld(out), _stream()))
    else:
        check(lib().sm_upsample_nearest(_ptr(x), _ptr(out), n, h, w, c, sy, sx, x_ld, _ld(out), _stream()))
    return out


def copy_channels(x, out):
    n, h, w, c, x_ld = _check_nhwc(x)
    check(lib().sm_copy_channels(_ptr(x), _ptr(out), n * h * w, c, x_ld, _ld(out), _stream()))
    return out


def reshape_nchw_order(x, c, h, w, out=None):
    """MPSNNReshapeNode for a reshape that is not an alias in NHWC: out[n, y, x, c] = element c*h*w + y*w + x of the source's (c, y, x) order"""
    n, hs, ws, cs, x_ld = _check_nhwc(x)
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=x.device) if out is None else out
    check(lib().sm_reshape_nchw_order(_ptr(x), _ptr(out), n, cs, hs * ws, x_ld, c, h * w, _ld(out), _stream()))
    return out


def nchw_f32_to_nhwc_f16(x, ld=None, halo=(0, 0, 0, 0)):
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    n, c, h, w = x.shape
    ld = c if ld is None else ld
    out = torch.empty((n, h + halo[0] + halo[2], w + halo[1] + halo[3], ld), dtype=torch.float16, device=x.device)
    check(lib().sm_nchw_f32_to_nhwc_f16(_ptr(x), _ptr(out), n, c, h, w, ld, halo[0], halo[1], halo[2], halo[3], _stream()))
    return out


def nhwc_f16_to_nchw_f32(x):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(lib().sm_nhwc_f16_to_nchw_f32(_ptr(x), _ptr(out), n, c, h, w, x_ld, _stream()))
    return out


def nhwc_f16_to_float_array(x):
    n, h, w, c, x_ld = _check_nhwc(x)
    out = torch.empty((lib().sm_float_array_count(n, c, h, w),), dtype=torch.float32, device=x.device)
    check(lib().sm_nhwc_f16_to_float_array(_ptr(x), _ptr(out), n, c, h, w, x_ld, _stream()))
    return out


def stem3x3(x_nchw, w_ohwi, bias=None, stride=1, pad=1, act=ACT_NONE, slope=0.0):
    """3x3 image stem straight from the fp32 NCHW batch (sm_stem3x3_f32_fwd): w_ohwi [Cout,3,3,C] fp16, C <= 3; returns NHWC fp16"""
    assert x_nchw.is_cuda and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and w_ohwi.dtype == torch.float16 and w_ohwi.is_contiguous()
    n, c, ih, iw = x_nchw.shape
    cout = w_ohwi.shape[0]
    assert tuple(w_ohwi.shape[1:]) == (3, 3, c)
    ho, wo = (ih + 2 * pad - 3) // stride + 1, (iw + 2 * pad - 3) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x_nchw.device)
    check(lib().sm_stem3x3_f32_fwd(_ptr(x_nchw), n, c, ih, iw, _ptr(w_ohwi), _ptr(bias), cout, stride, pad, act, slope, _ptr(y), _ld(y), _stream()))
    return y


def mbconv_stem(x_nchw, w_stem_ohwi, b_stem, wd, bd, w2, b2, st_stride=2, st_pad=1, stride=1, acts=(ACT_RELU, ACT_RELU, ACT_NONE)):
    """image stem (3x3 on the fp32 NCHW batch) -> depthwise 3x3 -> 1x1 project in one launch (sm_mbconv_stem_fwd); w_stem_ohwi [Ch,3,3,3] fp16"""
    from ._native import MbConvDesc
    assert x_nchw.is_cuda and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    n, c, ih, iw = x_nchw.shape
    ch, cout = wd.shape[0], w2.shape[0]
    h, w = (ih + 2 * st_pad - 3) // st_stride + 1, (iw + 2 * st_pad - 3) // st_stride + 1
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x_nchw.device)
    d = MbConvDesc(n, h, w, ho, wo, stride, 27, ch, cout, 1, 0, acts[0], 0.0, acts[1], 0.0, acts[2], 0.0, 0, _ld(y))
    check(lib().sm_mbconv_stem_fwd(C.byref(d), _ptr(x_nchw), c, ih, iw, st_stride, st_pad, _ptr(w_stem_ohwi), _ptr(b_stem), _ptr(wd), _ptr(bd), _ptr(w2), _ptr(b2),
                                   _ptr(y), _stream()))
    return y
