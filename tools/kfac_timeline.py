"""Scratch: per-stream timeline summary of ONE KFAC factor build from a rocprofv3 kernel trace db:
when does each stream finish, which kernels form the tail after the last autograd kernel?"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(con.execute(f"select name, start, end, {qcol or 0} from kernels order by start"))
# split into builds by large gaps (> 1 ms)
builds, cur = [], [rows[0]]
for r in rows[1:]:
    if r[1] - cur[-1][2] > 20_000_000:
        builds.append(cur); cur = [r]
    else:
        cur.append(r)
builds.append(cur)
print("columns:", cols)
builds = [b for b in builds if len(b) > 100]
print([ (len(x), round((max(r[2] for r in x)-x[0][1])/1e6,2)) for x in builds])
b = builds[-1]
t0, t1 = b[0][1], max(r[2] for r in b)
print(f"{len(builds)} builds; last: {len(b)} kernels, {(t1 - t0) / 1e6:.2f} ms")
by_q = collections.defaultdict(list)
for r in b:
    by_q[r[3]].append(r)
for q, rs in by_q.items():
    busy = sum(r[2] - r[1] for r in rs) / 1e6
    print(f"queue {q}: {len(rs)} kernels, busy {busy:.2f} ms, first {(rs[0][1]-t0)/1e6:.2f} ms, last end {(max(r[2] for r in rs)-t0)/1e6:.2f} ms")
clo = [r for r in b if "clo::" in r[0]]
oth = [r for r in b if "clo::" not in r[0]]
last_oth = max(r[2] for r in oth)
print(f"last non-clo kernel ends at {(last_oth - t0)/1e6:.2f} ms; tail of clo kernels after it:")
for r in clo:
    if r[2] > last_oth:
        print(f"   {(r[1]-t0)/1e6:7.3f} -> {(r[2]-t0)/1e6:7.3f} ms  {r[0][:90]}")
