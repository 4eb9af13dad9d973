"""Compressed view of a kernel's instruction stream from hipcc --save-temps output:
    python tools/asm_pattern.py file.s <kernel-name-substring> [max_chars]
M = v_mfma, r = ds_read, w = ds_write, D = LDS-DMA, L = global load, S = global store, v = VALU, s = SALU (omitted), W(..) = s_waitcnt, |BAR|"""
import re, sys
s = open(sys.argv[1]).read(); name = sys.argv[2]; mx = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
m = re.search(r"^(\S*%s\S*):\s*;[^\n]*\n(.*?)s_endpgm" % re.escape(name), s, re.S | re.M)
keys = []
for l in m.group(2).split("\n"):
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")): continue
    op = t[0]
    if op.endswith(":"): keys.append(" [%s] " % op[:-1]); continue
    if op.startswith("v_mfma"): keys.append("M")
    elif op.startswith("ds_read"): keys.append("r")
    elif op.startswith("ds_write"): keys.append("w")
    elif op.startswith("s_waitcnt"): keys.append(" W(" + " ".join(t[1:]) + ") ")
    elif "load_lds" in op or (op.startswith("buffer_load") and "lds" in l): keys.append("D")
    elif op.startswith(("global_load", "buffer_load")): keys.append("L")
    elif op.startswith(("global_store", "buffer_store")): keys.append("S")
    elif op.startswith("s_barrier"): keys.append(" |BAR| ")
    elif op.startswith("s_cbranch"): keys.append(" <br %s> " % t[-1])
    elif op.startswith("v_"): keys.append("v")
print("".join(keys)[:mx])
