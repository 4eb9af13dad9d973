"""Stamp the measured numbers into the docs FROM the committed profile files (three rounds of hand-copied figures had drifted from the files they cite).

    python tools/doc_numbers.py [--tag r05] [--check]

Every block between `<!-- numbers:NAME -->` and `<!-- /numbers:NAME -->` in DESIGN.md / README.md is REPLACED by text generated from profiles/<tag>_*:
    state    bench lines of the three single-GPU BASELINE configs (<tag>_bench{,_mbv2,_unet}.json) + the same-box A/B files
    kernel   rocprofv3 kernel-trace summary of the bench command (<tag>_kernel_stats.txt, last line) + the conv family line of the bench JSON
    pmc      HBM bytes per step by PMC against the two byte models (<tag>_pmc_traffic{,_mbv2,_unet}.json + the bench JSON's byte_models)
    t0       per-launch fixed cost (<tag>_fixed_cost_fit.json)
    parity   both norms at full size on the first and the last image of the batch (<tag>_parity_full_size.json)
    stage    per-phase timeline of the persistent runs (<tag>_stage_timeline.txt) and their in-graph A/B (<tag>_ab_stage_same_box.txt)
--check: exit 1 if a file would change (the CPU test suite runs it: a doc edited by hand inside a block, or a profile replaced without re-stamping, fails the test).
Nothing outside the markers is touched; prose that quotes a number belongs INSIDE a block or cites the file instead of the figure."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)


def load(name):
    path = P(name)
    return json.load(open(path)) if os.path.exists(path) else None


def last_json_line(name):
    path = P(name)
    if not os.path.exists(path):
        return None
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def ab_pairs(name, a_label, b_label):
    """(list of a ms, list of b ms) from a same-box A/B log of tools/gpu_probe.py lines"""
    path = P(name)
    if not os.path.exists(path):
        return {}
    out = {}
    for l in open(path):
        m = re.match(r"(.+?)\s+(r50|mbv2|unet) bs=\d+: ([\d.]+) ms/iter", l)
        if m:
            out.setdefault(m.group(2), {}).setdefault(m.group(1).strip(), []).append(float(m.group(3)))
    return out


def block_state(tag):
    items = []
    names = {"": ("ResNet-50 bs=64", "r50"), "_mbv2": ("MobileNet-v2 bs=256", "mbv2"), "_unet": ("U-Net 512² bs=16", "unet")}
    ab = ab_pairs("%s_ab_same_box.txt" % tag, "", "")
    for suf, (label, key) in names.items():
        b = last_json_line("%s_bench%s.json" % (tag, suf))
        if not b:
            continue
        r = b["roofline"]
        abtxt = "no same-box pair"
        if key in ab and len(ab[key]) == 2:
            (la, va), (lb, vb) = sorted(ab[key].items(), key=lambda kv: 0 if "round" in kv[0] or "binary" in kv[0] else 1)
            ma, mb = sum(va) / len(va), sum(vb) / len(vb)
            abtxt = "%s → %s ms (**%+.1f %%**)" % (" / ".join("%.3f" % v for v in va), " / ".join("%.3f" % v for v in vb), (mb / ma - 1) * 100)
        frac = "%.3f of the %s peak (%s %s)" % (r["frac"], "2.5 PFLOP/s MFMA" if r["bound"] == "mfma" else "8 TB/s HBM", r["achieved"], r["unit"])
        items.append("* **%s** — `bench.py`, driver form (`profiles/%s_bench%s.json`): **%.4f ms, %.1f k img/s**, %d launches, `roofline.frac` %s.  Previous round's binary → this tree, "
                     "same box, three alternating runs (`profiles/%s_ab_same_box.txt`): %s." % (label, tag, suf, b["ms_per_step"], b["value"] / 1e3, b["config"]["launches_per_step"], frac, tag, abtxt))
    sec = last_json_line("%s_bench.json" % tag)
    if sec:
        runs = sec["config"].get("persistent_runs")
        if runs is not None:
            items.append("* Persistent runs of that ResNet-50 line (`config.persistent_runs`): " + ("; ".join(
                "%d layers from step %d: one launch %.1f µs against %.1f µs layer by layer → %s" % (r["layers"], r["first_step"], r["t_run_us"], r["t_layers_us"], "on" if r["on"] else "off (" + ", ".join(r["why_off"]) + ")")
                for r in runs) or "none") + ".")
    if sec and "secondary" in sec and "relu6_clip_export" in sec["secondary"].get("mbv2", {}):
        q = sec["secondary"]["mbv2"]["relu6_clip_export"]
        items.append("* MobileNet-v2 as a torchvision export holds it (every activation an ONNX `Clip(0, 6)`): %.4f ms, %.1f k img/s, %d launches (`secondary.mbv2.relu6_clip_export` of the same record)." % (
            q["ms_per_step"], q["value"] / 1e3, q["launches_per_step"]))
    return "\n".join(items)


def block_kernel(tag):
    path = P("%s_kernel_stats.txt" % tag)
    b = last_json_line("%s_bench.json" % tag)
    out = []
    if os.path.exists(path):
        fam = [l for l in open(path) if l.startswith("# MFMA conv/FC family")]
        if fam:
            out.append("rocprofv3 `--kernel-trace --stats` of the bench command (`profiles/%s_kernel_stats.txt`): %s" % (tag, fam[-1][2:].strip()))
    if b:
        r = b["roofline"]
        out.append("`bench.py` of the same tree (`profiles/%s_bench.json`): %s, %.2f µs average, %.1f TFLOP/s = `roofline.frac` %.4f; whole step %.4f ms by HIP events." % (
            tag, r["kernel"], r["avg_launch_us"], r["achieved"], r["frac"], r["region_ms_per_step_hip_events"]))
    return "\n".join(out)


def block_pmc(tag):
    rows = []
    for suf, label in (("", "ResNet-50 bs 64"), ("_mbv2", "MobileNet-v2 bs 256"), ("_unet", "U-Net bs 16")):
        p, b = load("%s_pmc_traffic%s.json" % (tag, suf)), last_json_line("%s_bench%s.json" % (tag, suf))
        if not p or not b:
            continue
        bm = b["roofline"]["byte_models"]
        rd, wr = p["hbm_read_bytes_per_step"] / 1e9, p["hbm_write_bytes_per_step"] / 1e9
        rows.append("| %s | %.2f GB | %.2f GB | %.2f + %.2f = **%.2f GB** | %.2f× |" % (
            label, bm["graph_per_layer_bytes_per_step"] / 1e9, bm["post_fusion_compulsory_bytes_per_step"] / 1e9, rd, wr, rd + wr, (rd + wr) * 1e9 / bm["post_fusion_compulsory_bytes_per_step"]))
    return ("Bytes per step.  Per-layer: `smelter_exec_algorithmic_bytes` (every graph tensor written and read once).  Compulsory: `smelter_exec_launch_bytes` (what the launches must move "
            "after fusion).  PMC: `profiles/%s_pmc_traffic*.json`, FETCH_SIZE ×2 KiB + WRITE_SIZE KiB in separate passes (L2 ↔ fabric traffic).\n\n"
            "| model | per-layer | compulsory | PMC read + written | PMC ÷ compulsory |\n|---|---|---|---|---|\n" % tag) + "\n".join(rows)


def block_t0(tag):
    f = load("%s_fixed_cost_fit.json" % tag)
    if not f:
        return "(no %s_fixed_cost_fit.json)" % tag
    pts = ", ".join("bs %d: %.4f ms" % (p["batch"], p["ms_per_step_median"]) for p in f["points"])
    return ("`profiles/%s_fixed_cost_fit.json` (`tools/fixed_cost_fit.py`: %s; least squares `t = t0 + slope·bs`): **t0 = %.4f ms = %.2f µs × %d launches**, slope %.2f µs / image "
            "(%.0f TFLOP/s on the work term); the batch-independent share of the bs-64 step is %.0f %%." % (
                tag, pts, f["fit"]["t0_ms"], f["fixed_cost_us_per_launch"], f["launches_per_step"], f["fit"]["slope_ms_per_image"] * 1e3, f["work_rate_tflops"], f["fixed_share_at_largest_batch"] * 100))


def block_parity(tag):
    d = load("%s_parity_full_size.json" % tag)
    if not d:
        return "(no %s_parity_full_size.json)" % tag
    rows = []
    for k, label in (("resnet50", "ResNet-50 bs 64"), ("mobilenet_v2", "MobileNet-v2 bs 256"), ("unet", "U-Net bs 16")):
        if k not in d:
            continue
        e = d[k]; q = e["error_over_max_ref"]
        rows.append("| %s | %s | %.2e (%.2e) | %.2e (%.2e) | %d of %d |" % (label, e["checked_images"], e["rel_l2"], e["floor_rel_l2"], e["nerr_vs_fp16_storage_oracle"], e["floor_fp32_vs_fp64_accumulation"],
                                                                         q["count_above_1e-3_of_max_ref"], q["elements"]))
    return ("`profiles/%s_parity_full_size.json` (final binary; both norms against the fp16-storage oracle, in brackets the floor = the oracle's own fp32-vs-fp64 accumulation difference):\n\n"
            "| model | images checked | rel-L2 (floor) | max-norm (floor) | elements above 1e-3·max\\|ref\\| |\n|---|---|---|---|---|\n" % tag) + "\n".join(rows)


def block_stage(tag):
    out = []
    path = P("%s_stage_timeline.txt" % tag)
    if os.path.exists(path):
        cur, tables = None, {}
        for l in open(path):
            m = re.match(r"status \(\d+, (\d)\) kinds \[(.*)\]", l)
            if m:
                cur = m.group(2); tables[cur] = {"mode": m.group(1), "rows": [], "total": None}
            elif cur and re.match(r"\d+\s+\d+\s", l):
                tables[cur]["rows"].append(l.split("|")[1].split() + l.split("|")[0].split())
            elif cur and l.startswith("kernel:"):
                tables[cur]["total"] = float(l.split()[-2])
        for kinds, t in tables.items():
            k = [int(x) for x in kinds.split(",")]
            name = "7 × 7 × 2048 (dual head + 2 bottleneck blocks)" if set(k) <= {0, 1, 2, 6, 7, 8} else "14 × 14 × 1024 (dual head + 5 bottleneck blocks)"
            name += ", 8-wave tile forms (two waves per SIMD)" if min(k) >= 6 else ", 4-wave tile forms"
            out.append("ResNet-50 bs 64, %s, %d phases, ONE launch: **%.1f µs** first stamp → last stamp (`profiles/%s_stage_timeline.txt`; hand-off mode %s = XCD-local).  Per phase, µs, averages over 256 tiles:\n" % (
                name, len(k), t["total"], tag, t["mode"]))
            out.append("| phase | tile form | wait for producers | first patch fill | main loop | epilogue | store drain + publish |\n|---|---|---|---|---|---|---|")
            seen = set()
            for i, r in enumerate(t["rows"]):
                key = (k[i], i == 0)
                if key in seen:
                    continue
                seen.add(key)
                out.append("| %d%s | %d | %s | %s | %s | %s | %s |" % (i, " (head: second source)" if i == 0 else "", k[i], r[0], r[1], r[2], r[3], r[4]))
            out.append("")
    ab = P("%s_ab_stage_same_box.txt" % tag)
    if os.path.exists(ab):
        vals = {}
        for l in open(ab):
            m = re.match(r"(layers only|runs \(default\)|runs \+ 14x14)\s+r50 bs=64: ([\d.]+) ms/iter.*launches (\d+)", l)
            if m:
                vals.setdefault(m.group(1), []).append((float(m.group(2)), int(m.group(3))))
        if vals:
            fmt = lambda k: "%s ms (%d launches)" % (" / ".join("%.3f" % v for v, _ in vals[k]), vals[k][0][1]) if k in vals else "—"
            out.append("In the graph, same box, three alternating rounds (`profiles/%s_ab_stage_same_box.txt`): every layer its own launch %s; runs where the compile's in-situ A/B keeps them (default) %s; "
                       "with the 14 × 14 forms admitted (`SM_FUSE_STAGE_WIDE`) %s." % (tag, fmt("layers only"), fmt("runs (default)"), fmt("runs + 14x14")))
            for n in [l.strip() for l in open(ab) if "persistent run" in l][:4]:
                m = re.search(r"persistent run (?:at step \d+ \((\d+) layers\)|of (\d+) layers): ([\d.]+) us(?: as one launch)?, (?:layer by layer )?([\d.]+) us(?: layer by layer)? -> (\w+)", n)
                if m:
                    out.append("  The compile's own in-situ measurement: run of %s layers %s µs as one launch, %s µs layer by layer → kept as %s." % (m.group(1) or m.group(2), m.group(3), m.group(4), m.group(5)))
    return "\n".join(out)


def block_bneck(tag):
    out = []
    ab = P("%s_ab_bneck_same_box.txt" % tag)
    if os.path.exists(ab):
        vals = {}
        for l in open(ab):
            m = re.match(r"(blocks as reduce.*?|whole blocks.*?)\s+r50 bs=64: ([\d.]+) ms/iter.*launches (\d+)", l)
            if m:
                vals.setdefault(m.group(1).strip(), []).append((float(m.group(2)), int(m.group(3))))
        for k, v in vals.items():
            out.append("* %s: %s ms (%d launches)" % (k, " / ".join("%.3f" % t for t, _ in v), v[0][1]))
        if len(vals) == 2:
            (a, va), (b, vb) = list(vals.items())
            out.append("  Same box, three alternating rounds (`profiles/%s_ab_bneck_same_box.txt`): **%+.1f %%**." % (tag, (sum(t for t, _ in vb) / len(vb)) / (sum(t for t, _ in va) / len(va)) * 100 - 100))
    bb = P("%s_bneck_bench.txt" % tag)
    if os.path.exists(bb):
        cur = None
        for l in open(bb):
            if l.startswith("=="):
                cur = l.strip("= \n")
            m = re.match(r"chained: two launches ([\d.]+) us \((\d+) MB.*one launch ([\d.]+) us \((\d+) MB compulsory, ([\d.]+) TB/s, (\d+) TFLOP/s\)", l)
            if m and cur:
                out.append("* `%s` (`profiles/%s_bneck_bench.txt`; hipGraph of chained launches): the launches it replaces %s µs (%s MB), ONE launch **%s µs** (%s MB compulsory, %s TB/s, %s TFLOP/s)." % (
                    cur, tag, m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), m.group(6)))
    adm = P("%s_run_admission.txt" % tag)
    if os.path.exists(adm):
        lines = [l.strip() for l in open(adm)]
        vec = [re.search(r"on = (\[.*?\])", l).group(1) for l in lines if l.startswith("compile ")]
        same = [l for l in lines if l.startswith("same `on` vector")]
        runs = [l for l in lines if l.startswith("run at step")]
        if vec:
            out.append("* Admission of the persistent runs over %d compiles in one process (`profiles/%s_run_admission.txt`, `tools/run_admission.py`): `on` = %s — %s" % (
                len(vec), tag, " ".join(vec), same[0] if same else ""))
            ts = [re.search(r"one launch ([\d.]+) us \| layer by layer ([\d.]+) us", l) for l in runs]
            ts = [(float(m.group(1)), float(m.group(2))) for m in ts if m]
            if ts:
                out.append("  The A/B medians behind them: one launch %s µs, layer by layer %s µs." % (" / ".join("%.1f" % a for a, _ in ts), " / ".join("%.1f" % b for _, b in ts)))
    return "\n".join(out)


BLOCKS = {"bneck": block_bneck, "state": block_state, "kernel": block_kernel, "pmc": block_pmc, "t0": block_t0, "parity": block_parity, "stage": block_stage}


def wrap(text, width=150):
    """generated prose in lines of at most `width` columns (list items with a hanging indent); table rows stay whole"""
    import textwrap
    out = []
    for l in text.split("\n"):
        if len(l) <= width or l.lstrip().startswith("|"):
            out.append(l); continue
        m = re.match(r"^(\s*)([*-]\s+)?", l)
        first = m.group(0); rest = m.group(1) + " " * len(m.group(2) or "")
        out.extend(textwrap.wrap(l[len(first):], width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False))
    return "\n".join(out)


def stamp(text, tag):
    def rep(m):
        name = m.group(1)
        if name not in BLOCKS:
            return m.group(0)
        return "<!-- numbers:%s -->\n%s\n<!-- /numbers:%s -->" % (name, wrap(BLOCKS[name](tag).rstrip()), name)
    return re.sub(r"<!-- numbers:(\w+) -->.*?<!-- /numbers:\1 -->", rep, text, flags=re.S)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r06")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    changed = []
    for doc in ("DESIGN.md", "README.md"):
        path = os.path.join(ROOT, doc)
        old = open(path).read()
        new = stamp(old, a.tag)
        if new != old:
            changed.append(doc)
            if not a.check:
                open(path, "w").write(new)
    print(("would change: " if a.check else "stamped: ") + (", ".join(changed) or "nothing"))
    sys.exit(1 if a.check and changed else 0)


if __name__ == "__main__":
    main()
