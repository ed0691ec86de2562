"""Gaps > min_us in the last `window_s` seconds of a rocprofv3 kernel trace, with the kernels on either side (dev tool)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); min_us = float(sys.argv[2]); window = float(sys.argv[3])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
t_end = rows[-1][2]
sel = [r for r in rows if r[1] > t_end - window * 1e9]
gaps, prev = [], None
for n, s, e in sel:
    if prev and s - prev[2] > min_us * 1e3:
        gaps.append(((s - prev[2]) / 1e3, prev[0], n, (s - sel[0][1]) / 1e6))
    if prev is None or e > prev[2]:
        prev = (n, s, e)
sh = lambda n: n.replace("void ", "").replace("pulse::", "").replace("at::native::", "")[:46]
print(len(gaps), f"gaps > {min_us} us in the last {window} s; total {sum(g[0] for g in gaps) / 1e3:.1f} ms")
for g in sorted(gaps, key=lambda g: -g[0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 20]:
    print(f"{g[0]:9.1f} us at +{g[3]:7.1f} ms  {sh(g[1])} -> {sh(g[2])}")
