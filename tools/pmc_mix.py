"""Instruction mix per launch of the LAST encode() step, from rocprofv3 --pmc passes of bench.py (tools/pmc_mix.sh): how many scalar / vector / LDS / memory instructions a wave issues
per MFMA, the share of wave cycles spent waiting, MFMA duty.  python tools/pmc_mix.py gpurun_out/pmc_mix/p*/**/p_counter_collection.csv"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import step_starts  # noqa: E402

disp = collections.OrderedDict()
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    ids, seen = [], set()
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d not in seen:
            seen.add(d); ids.append((d, r["Kernel_Name"]))
    starts = step_starts([n for _, n in ids], 2)
    lo, hi = starts[-2], starts[-1]
    ordinal = {d: i - lo for i, (d, _) in enumerate(ids) if lo <= i < hi}
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d in ordinal:
            e = disp.setdefault(ordinal[d], {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "")})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("%3s %-46s %9s %7s %7s %7s %7s %7s | %6s %6s %6s" % ("#", "kernel", "mfma", "valu/m", "salu/m", "lds/m", "vmem/m", "all/m", "wait%", "issue%", "mfma%"))
tot = collections.Counter()
for i, e in disp.items():
    name = e["name"]
    key = next((k for k in ("stage_kernel", "conv_igemm", "conv3x3_strip", "conv3x3_ws", "conv3x3_p9", "conv3x3_kq", "conv1x1_ps", "conv1x1_rw", "conv_stem_pool", "mbconv", "pw_thin", "stem3") if k in name), None)
    short = name[name.find(key):][:44] if key else name[:44]
    m = e.get("SQ_INSTS_MFMA", 0); v = e.get("SQ_INSTS_VALU", 0) - m; s = e.get("SQ_INSTS_SALU", 0); l = e.get("SQ_INSTS_LDS", 0); vm = e.get("SQ_INSTS_VMEM_RD", 0) + e.get("SQ_INSTS_VMEM_WR", 0)
    wc = e.get("SQ_WAVE_CYCLES", 0)
    for k, x in (("m", m), ("v", v), ("s", s), ("l", l), ("vm", vm)): tot[k] += x
    r = lambda x: x / m if m else float("nan")
    print("%3d %-46s %9.3g %7.2f %7.2f %7.2f %7.2f %7.2f | %6.1f %6.1f %6.1f" % (i, short, m, r(v), r(s), r(l), r(vm), r(v + s + l + vm), 100 * e.get("SQ_WAIT_ANY", 0) / wc if wc else 0,
                                                                                 100 * e.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else 0, 100 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc) * 1.0 if wc else 0))
m = tot["m"]
print("step: per MFMA %.2f VALU, %.2f SALU, %.2f LDS, %.2f VMEM" % (tot["v"] / m, tot["s"] / m, tot["l"] / m, tot["vm"] / m))
