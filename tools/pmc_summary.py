"""Summarise rocprofv3 --pmc runs of bench.py (counter_collection.csv) per op of the LAST encode() step and for the
conv_igemm kernel family; writes the HBM traffic per launch that bench.py reports as roofline.traffic.

    python tools/pmc_summary.py gpurun_out/pmc_r01/pass*/p_counter_collection.csv --out profiles/r01_pmc_traffic.json

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is in KiB and reports HALF the bytes of a wide coalesced
stream -> hbm_read_bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE (KiB) is used as reported (uncalibrated in the guide)."""
import os
import sys
import argparse
import collections
import csv
import json


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import step_starts  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csvs", nargs="+")
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--head", default="unrecorded", help="git revision of the measured binary (recorded in the JSON)")
    a = ap.parse_args()
    # dispatch_id -> {counter: value}, kernel name
    disp = collections.OrderedDict()
    for path in a.csvs:
        rows = list(csv.DictReader(open(path)))
        # steady state = dispatches after the (len-steps)th nchw_to_nhwc; dispatch ids differ per pass, so align by ordinal
        ids = []
        seen = set()
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
                e = disp.setdefault(ordinal[d], {"name": r["Kernel_Name"]})
                e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    tot = collections.Counter()
    conv = collections.Counter()
    nconv = 0
    print("%3s %-44s %10s %10s %8s %9s" % ("#", "kernel", "rd_MB", "wr_MB", "L2hit%", "mfma_busy"))
    for i, e in disp.items():
        name = e["name"]
        key = next((k for k in ("conv_igemm", "conv3x3_strip", "conv3x3_ws", "conv3x3_p9", "conv1x1_ps", "conv1x1_rw", "conv_stem_pool", "mbconv", "pw_thin") if k in name), None)
        short = name[name.find(key):][:40] if key else name[:40]
        rd = e.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6
        wr = e.get("WRITE_SIZE", 0) * 1024 / 1e6
        hit, miss = e.get("TCC_HIT_sum", 0), e.get("TCC_MISS_sum", 0)
        hr = 100 * hit / (hit + miss) if hit + miss else float("nan")
        print("%3d %-44s %10.1f %10.1f %8.1f %9.3g" % (i, short, rd, wr, hr, e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
        tot["rd"] += rd; tot["wr"] += wr
        if key or "splitk" in name:
            conv["rd"] += rd; conv["wr"] += wr; nconv += 1
    print("step total: read %.1f MB, write %.1f MB ; conv family: read %.1f MB, write %.1f MB over %d launches" % (
        tot["rd"], tot["wr"], conv["rd"], conv["wr"], nconv))
    if a.out:
        json.dump({"hbm_bytes_per_launch": round((conv["rd"] + conv["wr"]) * 1e6 / max(nconv, 1)),
                   "hbm_read_bytes_per_step": round(tot["rd"] * 1e6), "hbm_write_bytes_per_step": round(tot["wr"] * 1e6),
                   "conv_launches_per_step": nconv, "head": a.head,
                   "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace; FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1"},
                  open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
