"""Do kernels of one encode() overlap in time?  python tools/overlap_check.py <rocprofv3 results .db>  (last complete step)"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
rows = [(int(s), int(e), n) for n, s, e in c.execute("select name, start, end from kernels order by start")]
starts = [i for i, r in enumerate(rows) if "nchw_to_" in r[2]]
seg = rows[starts[-2]:starts[-1]]
t0 = seg[0][0]; prev_end = 0
for s, e, n in seg:
    m = re.search(r"conv_igemm\w*<[^>]*>", n); nm = m.group(0) if m else re.sub(r"\(.*", "", n)[-40:]
    flag = "  <-- starts %.1f us BEFORE the previous kernel ended" % ((prev_end - s) / 1e3) if s < prev_end else ""
    print("%8.1f %8.1f  %6.1f us  %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, nm, flag))
    prev_end = max(prev_end, e)
print("step span %.1f us" % ((seg[-1][1] - t0) / 1e3))
