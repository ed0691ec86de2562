"""Idle-gap analysis of a rocprofv3 kernel trace (rocpd SQLite): where does the GPU wait for the host?
Usage: python tools/rocprof_gaps.py DB [min_gap_us]"""
import collections
import sqlite3
import sys


def short(n):
    n = n.replace("void ", "").replace("pulse::", "")
    for pre in ("at::native::",):
        n = n.replace(pre, "")
    return n[:70]


def main(db_path, min_gap_us=8.0):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    gaps = collections.defaultdict(lambda: [0, 0.0])
    total_gap = busy = 0.0
    prev_end, prev_name = None, None
    for name, s, e in rows:
        if prev_end is not None:
            g = (s - prev_end) / 1e3
            if g > min_gap_us:
                k = (short(prev_name), short(name))
                gaps[k][0] += 1
                gaps[k][1] += g
                total_gap += g
        busy += (e - s) / 1e3
        if prev_end is None or e > prev_end:
            prev_end, prev_name = e, name
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"span {span / 1e3:.1f} ms, kernel time {busy / 1e3:.1f} ms, gaps > {min_gap_us} us: {total_gap / 1e3:.1f} ms")
    for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{t / 1e3:8.2f} ms  n={n:5d}  avg {t / n:8.1f} us   {a}  ->  {b}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0)
