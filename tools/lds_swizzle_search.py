"""Which XOR key makes the fragment reads of the wave-split 3x3 kernels conflict-free?  A fragment's pixels are TW-pixel tile rows inside RS-slot patch rows
(128-byte slots, eight 16-byte chunks; chunk c of slot (r, cc) is stored at chunk position c ^ key(r, cc)).  ds_read_b128 serves a wave in four groups of 16
lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32: MI355X_MICROARCH.md, LDS table); a group costs one LDS cycle per distinct address on its
busiest bank.  Searches key = (a (cc / 2) + b r) & 7 and prints the average / worst cycles per group over every (fragment, tap).
    python tools/lds_swizzle_search.py [--tw 14] [--mfma 32|16]"""
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--tw", type=int, default=14)
ap.add_argument("--rs", type=int, default=16)
ap.add_argument("--pixels", type=int, default=224)
ap.add_argument("--mfma", type=int, default=32, help="32: v_mfma_32x32x16 (lane = pixel l % 32, chunk l / 32); 16: v_mfma_16x16x32 (pixel l % 16, chunk l / 16)")
a = ap.parse_args()
G = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G = G + [[x + 32 for x in g] for g in G]
P = a.mfma


def cost(key):
    tot = n = worst = 0
    for f in range((a.pixels + P - 1) // P):
        for tap in range(9):
            for g in G:
                quads = {}
                for l in g:
                    t = f * P + l % P
                    if t >= a.pixels:
                        t = 0
                    ty, tx = divmod(t, a.tw)
                    r, cc = ty + tap // 3, tx + tap % 3
                    pos = (l // P) ^ key(r, cc)
                    slot = r * a.rs + cc
                    quads.setdefault((slot & 1) * 8 + (pos & 7), set()).add((slot, pos & 7))     # identical addresses broadcast: distinct ones queue
                m = max(len(v) for v in quads.values()); tot += m; n += 1; worst = max(worst, m)
    return tot / n, worst


res = sorted((cost(lambda r, cc, x=x, y=y: (x * (cc >> 1) + y * r) & 7), x, y) for x in range(8) for y in range(8))
print("usual key (slot / 2) & 7 = (a, b) = (1, %d): %s" % ((a.rs // 2) & 7, cost(lambda r, cc: ((r * a.rs + cc) >> 1) & 7)))
for c, x, y in res[:6]:
    print("key = (%d (cc/2) + %d r) & 7: avg %.3f worst %d cycles per group (1 = conflict-free)" % (x, y, c[0], c[1]))
