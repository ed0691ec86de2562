"""Turn a rocprofv3 (ROCm 7.2, rocpd SQLite output of `--kernel-trace --stats`) database into the
per-kernel summary table committed under profiles/.  Usage: python tools/rocprof_summary.py DB [OUT.md]"""
import sqlite3
import sys


def main(db_path, out=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % of kernel time |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")
    lines.append("")
    lines.append(f"kernel time total {total / 1e6:.2f} ms over a {((span[1] - span[0]) / 1e6):.2f} ms first-to-last-kernel span "
                 f"({len(rows)} distinct kernels, {sum(r[1] for r in rows)} dispatches)")
    text = "\n".join(lines)
    if out:
        with open(out, "a") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
