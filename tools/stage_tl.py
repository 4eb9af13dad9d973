"""Per-phase timeline of ONE sm_stage_fwd run (diagnostic build: tools/stage_tl.sh, SMELTER_LIB_PATH=build/libsmelter_hip_stagetl.so).
Every block stamps the 100 MHz wall clock (the same counter on every XCD) per tile: 0 tile start (weights requested), 1 producers published + block synchronised,
2 first patch landed, 3 main loop done, 4 epilogue done (stores issued), 5 stores drained + published.  Printed per phase: when the first / last tile of the phase starts and
ends (us after the first stamp of the kernel), and the average length of each segment.  Usage: python tools/stage_tl.py [stage4|stage3] [4w] [nopack]   (4w: the 4-wave tile forms only)
       python tools/stage_tl.py formK:SHAPE   four independent phases of one shape on tile form K (expand7 | reduce7 | conv7 | expand14 | reduce14 | conv14)"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from smelter_amd import ops
from smelter_amd._native import lib
from stage_bench import build


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "stage4"
    if which.startswith("form"):
        # formK:SHAPE — four INDEPENDENT phases of one shape on tile form K (no hand-off between them: the columns are the tile body alone): expand7 | reduce7 | conv7 | expand14 | reduce14 | conv14
        kind, shape = int(which[4:].split(":")[0]), which.split(":")[1]
        hw, cin, cout, k, res = {"expand7": (7, 512, 2048, 1, True), "reduce7": (7, 2048, 512, 1, False), "conv7": (7, 512, 512, 3, False),
                                 "expand14": (14, 256, 1024, 1, True), "reduce14": (14, 1024, 256, 1, False), "conv14": (14, 256, 256, 3, False)}[shape]
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
        st = ops.Stage()
        x = rn(64, hw, hw, cin).abs().half(); r = rn(64, hw, hw, cout).abs().half() if res else None
        for _ in range(4):
            st.add(x, (rn(cout, k, k, cin) * (2.0 / (k * k * cin)) ** 0.5).half(), (rn(cout) * 0.1).float(), residual=r, act=1, kind=kind)
        st.pack()
    else:
        hw, cin, mid, blocks = (7, 2048, 512, 2) if which == "stage4" else (14, 1024, 256, 5)
        st, layers = build(64, hw, cin, mid, blocks, pack="nopack" not in sys.argv, four_waves="4w" in sys.argv)
    L = lib()
    L.sm_stage_timeline_read.argtypes = [C.c_void_p, C.c_void_p]; L.sm_stage_timeline_read.restype = C.c_int
    for _ in range(5):
        st.run()
    torch.cuda.synchronize()
    tl = np.zeros((256, 64, 8), np.uint64); cnt = np.zeros(256, np.uint32)
    assert L.sm_stage_timeline_read(tl.ctypes.data, cnt.ctypes.data) == 0
    print("status", st.status(), "kinds", st.kinds())
    rows = []
    for b in range(256):
        for t in range(int(cnt[b])):
            e = tl[b, t]; tag = int(e[7]); rows.append((tag >> 32 & 0xffff, tag >> 48, tag & 0xffffffff, b, [int(v) for v in e[:6]]))
    t0 = min(r[4][0] for r in rows)
    us = lambda v: (v - t0) / 100.0
    print("%-5s %-5s %9s %9s %9s | %8s %8s %8s %8s %8s  (us; averages over the phase's tiles)" % ("phase", "tiles", "first in", "last in", "last out", "wait", "1st fill", "main", "epilogue", "drain"))
    for ph in sorted(set(r[0] for r in rows)):
        rs = [r[4] for r in rows if r[0] == ph]
        seg = lambda a, b: np.mean([(r[b] - r[a]) / 100.0 for r in rs])
        print("%-5d %-5d %9.2f %9.2f %9.2f | %8.2f %8.2f %8.2f %8.2f %8.2f" % (ph, len(rs), us(min(r[0] for r in rs)), us(max(r[0] for r in rs)), us(max(r[5] for r in rs)),
                                                                               seg(0, 1), seg(1, 2), seg(2, 3), seg(3, 4), seg(4, 5)))
    print("kernel: first stamp -> last stamp %.2f us" % us(max(r[4][5] for r in rows)))
    if "starts" in sys.argv:                               # when does each block start its first tile (us after the earliest): the launch ramp + the mode decision
        st0 = sorted((us(r[4][0]), r[3], r[1]) for r in rows if r[0] == 0)
        print("first-tile starts, sorted:", " ".join("%.1f" % v[0] for v in st0[::8]))
        print("latest 8 (us, block, team):", st0[-8:])
    # one team in detail: team 0's blocks, start and end of each tile
    if "detail" in sys.argv:
        for r in sorted((r for r in rows if r[1] == 0), key=lambda r: (r[3], r[4][0])):
            print("team0 block %3d phase %2d tile %3d:" % (r[3], r[0], r[2]), " ".join("%7.2f" % us(v) for v in r[4]))


if __name__ == "__main__":
    main()
