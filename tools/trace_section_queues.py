"""Marker-bracketed section of a rocprofv3 kernel trace: kernels, kernel time, wall, and how the launches spread over the
hardware queues / streams the trace records.  usage: trace_section_queues.py results.db"""
import collections, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
sel = f"select name, start, end, grid_x{', ' + qcol if qcol else ''} from kernels order by start"
rows = list(con.execute(sel))
marks = [i for i, r in enumerate(rows) if "axpby" in r[0] and r[3] in (4352, 4099)]
a, b = marks[-2], marks[-1]
sec = rows[a + 1:b]
wall = (rows[b][1] - rows[a][2]) / 1e6
ktime = sum(r[2] - r[1] for r in sec) / 1e6
print(f"columns: {cols}")
print(f"section: {len(sec)} kernels, wall {wall:.3f} ms, kernel time {ktime:.3f} ms")
if qcol:
    byq = collections.defaultdict(lambda: [0, 0.0])
    for r in sec:
        byq[r[4]][0] += 1; byq[r[4]][1] += (r[2] - r[1]) / 1e6
    for q, (n, t) in sorted(byq.items(), key=lambda kv: -kv[1][1]):
        print(f"  {qcol} {q}: {n} kernels, {t:.3f} ms")
# idle time of the busiest queue: gaps between consecutive kernels of the whole section (any queue)
ev = sorted((r[1], r[2]) for r in sec)
busy_end, idle = ev[0][1], 0.0
for s, e in ev[1:]:
    if s > busy_end: idle += (s - busy_end) / 1e6
    busy_end = max(busy_end, e)
print(f"  time with NO kernel running: {idle:.3f} ms of {wall:.3f}")
