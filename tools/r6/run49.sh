R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prk
rocprofv3 --kernel-trace --stats -d /tmp/prk -o k -- python $R/tools/probe_cols.py 32 > /tmp/probe_k.log 2>&1
grep "K=" /tmp/probe_k.log
python - <<PY
import sqlite3
con=sqlite3.connect("/tmp/prk/k_results.db")
cur=con.execute("select * from kernels limit 1"); cols=[d[0] for d in cur.description]
seen={}
for row in con.execute("select * from kernels"):
    r=dict(zip(cols,row)); name=r['name']
    k=name.split('(')[0][-60:]
    if "clo::" in name and "mega" not in name:
        key=(k, r['grid_x'], r['grid_y'])
        d=seen.setdefault(key,[0,0.0,r]); d[0]+=1; d[1]+=r['duration']
for (k,gx,gy),(c,t,r) in sorted(seen.items(), key=lambda kv:-kv[1][1]):
    wg=r['workgroup_x']; blocks=gx//wg*max(1,gy//max(1,r['workgroup_y']))
    print(f"{k:62s} x{c:3d} avg {t/c/1e3:7.2f} us  blocks {blocks:6d} x {wg:4d} thr  lds {r['lds_size']:6d}  vgpr {r['vgpr_count']:3d}")
PY
