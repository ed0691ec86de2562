"""List the kernels between the k-th and (k+1)-th launch of a marker kernel (one rollout step). Usage: DB marker k"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker, k = sys.argv[2], int(sys.argv[3])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
for name, s, e in rows[a:b + 1]:
    n = name.replace("void ", "").replace("at::native::", "").replace("pulse::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n[:120]}")
