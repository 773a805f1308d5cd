"""Longest kernels and largest idle gaps of a rocprofv3 kernel trace (.db)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel" in t.lower())
cols = [r[1] for r in con.execute(f"pragma table_info({view})")]
print("view", view, cols)
rows = list(con.execute(f"select name, start, end, stream_id from {view} order by start")) if "stream_id" in cols else \
       [(*r, 0) for r in con.execute(f"select name, start, end from {view} order by start")]
print(len(rows), "kernels")
top = sorted(rows, key=lambda r: r[1] - r[2])[:8]
for n, s, e, q in top:
    print(f"long  {1e-3*(e-s):12.1f} us  stream {q}  {n[:90]}")
gaps = []
last_end = rows[0][2]
for i in range(1, len(rows)):
    n, s, e, q = rows[i]
    if s - last_end > 0:
        gaps.append((s - last_end, rows[i-1][0][:60], n[:60]))
    last_end = max(last_end, e)
for g, a, b in sorted(gaps, reverse=True)[:8]:
    print(f"gap   {1e-3*g:12.1f} us  after {a}  before {b}")
