R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pkf -o k -- python $R/tools/prof_kfac.py resnet > /dev/null 2>&1
python - <<'PY'
import sqlite3
con = sqlite3.connect("/tmp/pkf/k_results.db")
rows = list(con.execute("select name, start, end from kernels order by start"))
# last third = the third (steady-state) build
t_end = rows[-1][2]; 
import collections
# find builds by splitting on time gaps: simply take last 1/3 of kernels by count
sel = rows[2*len(rows)//3:]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in sel:
    k = n.split('(')[0].replace('void ', '')[:70]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"last build: {len(sel)} kernels, {tot/1e3:.2f} ms of kernel time, wall {(sel[-1][2]-sel[0][1])/1e6:.2f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{c:5d} {t:9.1f} us  {k}")
PY
