"""Summarise the marker-bracketed section(s) of a rocprofv3 kernel trace of tools/prof_kfac_build.py.
usage: kfac_trace_summary.py results.db rows"""
import collections, sqlite3, sys

con = sqlite3.connect(sys.argv[1])
rows_b = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rows = list(con.execute("select name, start, end, grid_x from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "axpby" in r[0] and r[3] in (4352, 4099, 4096 + 256)]
if len(marks) < 2:  # fall back: any axpby launches
    marks = [i for i, r in enumerate(rows) if "axpby" in r[0]][-3:]
import os
sections = [(os.environ.get("SECTION_TITLE", "factor build"), marks[0], marks[1])] + ([("damped Cholesky inverses", marks[1], marks[2])] if len(marks) > 2 else [])
for title, a, b in sections:
    sel = rows[a + 1:b]
    wall = (rows[b][1] - rows[a][2]) / 1e3
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e, g in sel:
        k = n.split("(")[0].replace("void ", "")[:86]
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    clo = sum(v[1] for k, v in agg.items() if "clo::" in k)
    print(f"== {title}: {len(sel)} kernels, wall {wall / 1e3:.3f} ms between the markers, kernel time {tot / 1e3:.3f} ms "
          f"(clo:: {clo / 1e3:.3f} ms, framework {(tot - clo) / 1e3:.3f} ms; streams overlap)")
    print("#  calls   total_us    avg_us  kernel")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{c:7d} {t:10.1f} {t / c:9.2f}  {k}")
    if a == marks[0]:
        print(f"clo_kernel_us {clo:.1f}")
