"""Summarise a rocprofv3 --kernel-trace run of bench.py (rocpd sqlite or kernel_trace.csv) into the text files
committed under profiles/.  Steady-state steps are delimited by the input-conversion kernel (first launch of every
encode); autotuning / warm-up dispatches before the last `--steps` segments are ignored.

    python tools/rocprof_summary.py gpurun_out/prof_x/r_results.db --steps 20 > profiles/r01_kernel_stats.txt
"""
import argparse
import collections
import csv
import re
import sqlite3
import sys


def load(path):
    rows = []  # (start_ns, end_ns, name)
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        for name, s, e in c.execute("select name, start, end from kernels order by start"):
            rows.append((int(s), int(e), name))
    else:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        rows.sort()
    return rows


def short(name):
    m = re.search(r"(stage_kernel|conv_igemm\w*|conv3x3_strip_kernel|conv3x3_wsx_kernel|conv3x3_ws_kernel|conv3x3_p9_kernel|conv1x1_ps_kernel|conv1x1_rwl?_kernel|conv_stem_pool_kernel|mbconv\w*|pw_thin_kernel)<[^>]*>", name)
    if m:
        return m.group(0)
    m = re.search(r"N_1\d+(\w+?)E", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if name.startswith("_ZN"):
        m = re.search(r"\d+([a-z_0-9]+_kernel)", name)
        if m:
            return m.group(1)
    return name


def step_starts(names, steps):
    """dispatch indices that open an encode() step.  Marker = the input-conversion kernel when the step has one; otherwise (the stem's loader
    waves read the fp32 batch) any kernel that occurs once per step: its last steps + 1 occurrences are evenly spaced"""
    starts = [i for i, n in enumerate(names) if "nchw_to_" in n]
    if len(starts) >= steps + 1:
        return starts
    occ = collections.defaultdict(list)
    for i, n in enumerate(names):
        occ[n].append(i)
    best, best_gap = [], 0
    for n, idx in occ.items():
        if len(idx) < steps + 1:
            continue
        tail = idx[-(steps + 1):]
        gaps = {y - x for x, y in zip(tail, tail[1:])}
        if len(gaps) == 1:
            g = gaps.pop()
            if g > best_gap or (g == best_gap and tail[0] < best[0]):
                best, best_gap = tail, g
    return best


# bs x 2 x MACs per image of the zoo models (smelter_amd/zoo.py return_macs; BASELINE.json batch sizes)
FLOPS_PER_STEP = {"r50": 64 * 2 * 4089184256.0, "mbv2": 256 * 2 * 300774272.0, "unet": 16 * 2 * 65523417088.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--steps", type=int, default=20)
    # algorithmic FLOPs of one step (2 * MACs of conv + fc) — per MODEL: the ResNet-50 default once put "496 TFLOP/s" under MobileNet-v2's table
    ap.add_argument("--model", choices=sorted(FLOPS_PER_STEP), default="r50")
    ap.add_argument("--flops-per-step", type=float, default=0.0, help="overrides --model")
    a = ap.parse_args()
    if not a.flops_per_step:
        a.flops_per_step = FLOPS_PER_STEP[a.model]
    rows = load(a.path)
    starts = step_starts([r[2] for r in rows], a.steps)
    assert len(starts) >= a.steps + 1, "found %d encode() calls" % len(starts)
    lo = starts[-a.steps - 1]
    hi = starts[-1]                      # the last encode may be followed by read-back kernels: use complete segments only
    seg = rows[lo:hi]
    span_ms = (rows[hi][0] - rows[lo][0]) / 1e6
    per = collections.OrderedDict()
    for s, e, n in seg:
        k = short(n)
        d = per.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
    total_us = sum(v[1] for v in per.values())
    print("# rocprofv3 --kernel-trace, steady state: last %d encode() steps, %d dispatches" % (a.steps, len(seg)))
    print("# wall span %.3f ms/step ; sum of kernel durations %.3f ms/step ; %d launches/step" % (span_ms / a.steps, total_us / 1e3 / a.steps, len(seg) // a.steps))
    print("%-46s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-46s %8d %12.1f %10.2f %6.1f%%" % (k, c, t, t / c, 100 * t / total_us))
    conv = [(c, t) for k, (c, t) in per.items() if k.startswith(("stage_kernel", "conv_igemm", "conv3x3_strip", "conv3x3_ws", "conv3x3_p9", "conv1x1_ps", "conv1x1_rw", "conv3x3_kq", "conv_stem_pool", "mbconv", "pw_thin")) or "splitk" in k]
    cc, ct = sum(c for c, _ in conv), sum(t for _, t in conv)
    print("# MFMA conv/FC family (stage_kernel = a persistent run of layers, conv_igemm*, conv3x3_strip/ws/wsx/p9, conv3x3_kq, conv1x1_ps / rw, split-K reduce): %d launches/step, avg %.2f us/launch, %.3f ms/step -> %.1f TFLOP/s (%.1f%% of 2500)" % (
        cc // a.steps, ct / cc, ct / 1e3 / a.steps, a.flops_per_step / (ct / a.steps * 1e-6) / 1e12, a.flops_per_step / (ct / a.steps * 1e-6) / 1e12 / 25))


if __name__ == "__main__":
    main()


def last_step_listing(path, steps):
    """per-dispatch durations of the last complete encode() step, in launch order"""
    rows = load(path)
    starts = [i for i, r in enumerate(rows) if "nchw_to_" in r[2]]
    lo, hi = starts[-2], starts[-1]
    out = []
    for s, e, n in rows[lo:hi]:
        out.append(((e - s) / 1e3, short(n)))
    return out
