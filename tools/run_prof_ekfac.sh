R=$PWD; python tools/prof_ekfac.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pe -o k -- python $R/tools/prof_ekfac.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, collections
con = sqlite3.connect("/tmp/pe/k_results.db")
rows = list(con.execute("select name, start, end from kernels order by start"))
sel = rows[-len(rows)//8:]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in sel:
    k = n.split('(')[0].replace('void ', '')[:90]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
print(f"tail: {len(sel)} kernels, {sum(v[1] for v in agg.values())/1e3:.2f} ms kernel time, wall {(sel[-1][2]-sel[0][1])/1e6:.2f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{c:5d} {t:9.1f} us  {k}")
PY
