"""Host side of the persistent run (smelter_amd/csrc/kernels/stage.hip: stage_plan, sm_stage_pack_weights) without a GPU: which tile form a phase gets on the ResNet-50 stage
shapes, what the planner refuses, and the fragment-order weight layout against a numpy restatement of the layout documented at sm_stage_pack_weights.  Addresses are stand-ins
(nothing is dereferenced by these entry points except the two host buffers of sm_stage_pack_weights).  The launch itself is tests/test_gpu_stage.py."""
import ctypes as C

import numpy as np
import pytest

from smelter_amd._native import ConvDesc, StagePhase, lib

ACT_RELU = 1
#            KG NG NF  NFR TAPS WCH   (stage.hip: StR7, StT7, StE7, StR14, StE14, StT14)
KINDS = {0: (4, 1, 4, 7, 1, 64), 1: (4, 1, 4, 7, 9, 64), 2: (1, 4, 4, 7, 1, 128), 3: (2, 2, 4, 7, 1, 128), 4: (1, 4, 4, 13, 1, 128), 5: (4, 1, 4, 13, 9, 64),
         6: (1, 8, 2, 7, 1, 128), 7: (4, 2, 2, 7, 1, 64), 8: (4, 2, 2, 7, 9, 64)}     # 6 .. 8: eight waves (two per SIMD)
XCH = {2, 4, 6}                       # forms whose tile leaves from the registers: weight rows assigned so that a lane holds eight consecutive couts


class Run:
    """phase list over a fake address space: every tensor gets its own 16-byte aligned range"""

    def __init__(self, n):
        self.n, self.next, self.ph = n, 0x100000000, []

    def tensor(self, h, w, c):
        a = self.next
        self.next += (self.n * h * w * c * 2 + 255) // 256 * 256
        return (a, h, w, c)

    def weights(self, cout, k, cin):
        a = self.next
        self.next += (cout * k * k * cin * 2 + 255) // 256 * 256
        return a

    def add(self, x, cout, k=1, res=None, x2=None, stride2=1, y=None, form=0, stride=1, w_addr=None):
        xa, h, w, k1 = x
        k2 = x2[3] if x2 else 0
        y = y or self.tensor(h, w, cout)
        p = StagePhase()
        pad = (k - 1) // 2
        p.d = ConvDesc(self.n, h, w, k1 + k2, cout, k, k, stride, stride, 1, 1, pad, pad, pad, pad, 1, h // stride, w // stride, ACT_RELU, 0.0, k1, cout, cout if res else 0)
        p.x, p.w_ohwi, p.bias, p.residual, p.y = xa, w_addr or self.weights(cout, k, k1 + k2), self.weights(cout, 1, 2), res[0] if res else None, y[0]
        if x2:
            p.x2, p.K2, p.x2_ld, p.x2_H, p.x2_W, p.x2_stride = x2[0], k2, k2, x2[1], x2[2], stride2
        p.form = form
        self.ph.append(p)
        return y

    def arr(self):
        return (StagePhase * len(self.ph))(*self.ph), len(self.ph)

    def supported(self):
        a, n = self.arr()
        return bool(lib().sm_stage_supported(a, n))

    def kinds(self):
        a, n = self.arr()
        return [lib().sm_stage_phase_kind(a, n, i) for i in range(n)]


def resnet_stage(n, hw, cin, mid, blocks, form=0):
    r = Run(n)
    t0, xin = r.tensor(hw, hw, mid), r.tensor(2 * hw, 2 * hw, cin // 2)
    cur = r.add(t0, cin, 1, x2=xin, stride2=2, form=form)             # expand + projection shortcut, concatenated along K (executor pass 3h)
    for _ in range(blocks):
        a = r.add(cur, mid, 1, form=form)
        b = r.add(a, mid, 3, form=form)
        cur = r.add(b, cin, 1, res=cur, form=form)
    return r


def test_resnet50_stage_shapes_get_their_tile_forms():
    s4, s3 = resnet_stage(64, 7, 2048, 512, 2), resnet_stage(64, 14, 1024, 256, 5)
    assert s4.supported() and s4.kinds() == [6] + [7, 8, 6] * 2          # every phase has an 8-wave form (two waves per SIMD): the run takes them
    s4f = resnet_stage(64, 7, 2048, 512, 2, form=-1)
    assert s4f.supported() and s4f.kinds() == [2] + [0, 1, 2] * 2         # form < 0: the 4-wave forms
    assert s3.supported() and s3.kinds() == [4] + [3, 5, 4] * 5          # 14 x 14: the 8-wave forms would hold half an image where forms 4 / 5 hold a whole one: the planner stays on four waves
    s38 = Run(64)                                                         # ... but they take the stage when named: half-image tiles (1x1), bands of seven rows (3x3)
    t0, xin = s38.tensor(14, 14, 256), s38.tensor(28, 28, 512)
    cur = s38.add(t0, 1024, 1, x2=xin, stride2=2, form=7)
    for _ in range(2):
        cur = s38.add(s38.add(s38.add(cur, 256, 1, form=8), 256, 3, form=9), 1024, 1, res=cur, form=7)
    assert s38.supported() and s38.kinds() == [6] + [7, 8, 6] * 2
    a, n = s4.arr()
    # control block: 48 header words + 18 phases x (batch rounded up to 16) counters
    assert lib().sm_stage_ctl_bytes(a, n) == (48 + 18 * 64) * 4
    # fragment-order copy: couts rounded up to the form's cout tile, every K
    for i, (cout, k) in enumerate([(2048, 1536), (512, 2048), (512, 4608), (2048, 512)]):
        assert lib().sm_stage_packed_weight_bytes(a, n, i) == cout * k * 2


def test_what_the_planner_refuses():
    ok = resnet_stage(8, 7, 2048, 512, 1)
    assert ok.supported()
    # more phases than the argument block holds
    many = resnet_stage(8, 7, 2048, 512, 6)
    assert len(many.ph) == 19 and not many.supported()
    # a stride-2 convolution is not a phase
    r = Run(8); x = r.tensor(14, 14, 256); r.add(x, 256, 3, stride=2)
    assert not r.supported()
    # in place
    r = Run(8); x = r.tensor(7, 7, 512); r.add(x, 512, 1, y=x)
    assert not r.supported()
    # residual operand = output
    r = Run(8); x = r.tensor(7, 7, 512); y = r.tensor(7, 7, 2048); r.add(x, 2048, 1, res=y, y=y)
    assert not r.supported()
    # two phases writing the same tensor
    r = Run(8); x = r.tensor(7, 7, 512); y = r.add(x, 512, 1); r.add(x, 512, 1, y=y)
    assert not r.supported()
    # a later phase overwrites what an earlier one reads
    r = Run(8); x = r.tensor(7, 7, 512); y = r.add(x, 512, 1); r.add(y, 512, 1, y=x)
    assert not r.supported()
    # unaligned operand
    r = Run(8); x = r.tensor(7, 7, 512); r.add((x[0] + 8, 7, 7, 512), 512, 1)
    assert not r.supported()
    # channel counts the tile forms do not divide
    r = Run(8); x = r.tensor(7, 7, 96); r.add(x, 512, 1)
    assert not r.supported()
    # a 3x3 image whose rows are longer than a tile (a tile is whole images or a band of whole rows)
    r = Run(2); x = r.tensor(120, 120, 256); r.add(x, 256, 3)
    assert not r.supported()
    r = Run(8); x = r.tensor(28, 28, 256); r.add(x, 256, 3)          # bands of two rows
    assert r.supported()
    # a 5 x 5 kernel
    r = Run(8); x = r.tensor(7, 7, 256); r.add(x, 256, 5)
    assert not r.supported()
    # a form that does not take the phase
    r = Run(8); x = r.tensor(7, 7, 512); r.add(x, 512, 3, form=1)        # form 0 is a 1x1 form
    assert not r.supported()
    assert lib().sm_stage_supported(None, 0) == 0


def pack_reference(w, kind, cout, k1, k2, nt):
    """[cout tile][ng][kg][super-block][unit = (tap, 32-channel piece)][fragment j][lane (l15, g)][8 halfs] (stage.hip, sm_stage_pack_weights)"""
    KG, NG, NF, _, TAPS, WCH = KINDS[kind]
    QPW, SBCH, BN = WCH // 32, KG * WCH, NG * NF * 16
    UPS, nsb = TAPS * QPW, (k1 + k2) // SBCH
    K = TAPS * k1 + k2
    w2 = w.reshape(cout, K)
    out = np.zeros((nt, NG, KG, nsb, UPS, NF, 64, 8), np.uint16)
    for nti in range(nt):
        for ng in range(NG):
            for kg in range(KG):
                for sb in range(nsb):
                    for ul in range(UPS):
                        for j in range(NF):
                            for lane in range(64):
                                l15, g = lane & 15, lane >> 4
                                co0 = nti * BN + ng * NF * 16
                                co = co0 + (j >> 1) * 32 + (l15 >> 2) * 8 + (j & 1) * 4 + (l15 & 3) if kind in XCH else co0 + j * 16 + l15
                                k = (ul // QPW) * k1 + sb * SBCH + kg * WCH + (ul % QPW) * 32 + g * 8
                                if co < cout:
                                    out[nti, ng, kg, sb, ul, j, lane] = w2[co, k:k + 8]
    return out.reshape(-1)


@pytest.mark.parametrize("kind,hw,cin,cout,k", [(0, 7, 512, 72, 1), (1, 7, 256, 64, 3), (2, 7, 128, 264, 1), (3, 14, 256, 136, 1), (4, 14, 128, 256, 1), (5, 14, 256, 40, 3), (6, 7, 512, 264, 1), (7, 7, 512, 72, 1), (8, 7, 256, 64, 3)])
def test_packed_weights_are_the_documented_fragment_order(kind, hw, cin, cout, k):
    r = Run(8)
    x = r.tensor(hw, hw, cin)
    r.add(x, cout, k, form=kind + 1)
    assert r.kinds() == [kind]
    a, n = r.arr()
    KG, NG, NF = KINDS[kind][:3]
    bn = NG * NF * 16
    nt = -(-cout // bn)
    nbytes = lib().sm_stage_packed_weight_bytes(a, n, 0)
    assert nbytes == nt * bn * k * k * cin * 2                           # ragged cout tiles are padded with zero rows
    rng = np.random.default_rng(kind)
    w = rng.integers(1, 65535, size=(cout, k, k, cin), dtype=np.uint16)   # bit patterns: the packer copies, it does not convert
    dst = np.zeros(nbytes // 2, np.uint16)
    assert lib().sm_stage_pack_weights(a, n, 0, w.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p)) == 0
    ref = pack_reference(w, kind, cout, cin, 0, nt)
    assert np.array_equal(dst, ref)
    # every weight appears exactly once; the rest is padding
    assert np.count_nonzero(dst) == w.size


def test_packed_weights_of_a_two_source_phase_keep_the_concatenated_k_order():
    r = Run(8)
    t0, xin = r.tensor(7, 7, 128), r.tensor(14, 14, 256)
    r.add(t0, 256, 1, x2=xin, stride2=2, form=3)
    assert r.kinds() == [2]
    a, n = r.arr()
    w = np.arange(256 * 384, dtype=np.uint32).astype(np.uint16).reshape(256, 1, 1, 384) | 1
    dst = np.zeros(lib().sm_stage_packed_weight_bytes(a, n, 0) // 2, np.uint16)
    assert lib().sm_stage_pack_weights(a, n, 0, w.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(dst, pack_reference(w, 2, 256, 128, 256, 1))
