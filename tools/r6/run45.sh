R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 65 160 256; do
rm -rf /tmp/pr$n
rocprofv3 --kernel-trace --stats -d /tmp/pr$n -o k -- python $R/tools/probe_c2.py $n > /tmp/probe_$n.log 2>&1
grep "N=" /tmp/probe_$n.log
python - <<PY
import sqlite3
con=sqlite3.connect("/tmp/pr$n/k_results.db")
cur=con.execute("select * from kernels limit 1"); cols=[d[0] for d in cur.description]
seen={}
for row in con.execute("select * from kernels"):
    r=dict(zip(cols,row)); name=r['name']
    k=name.split('(')[0][-44:]
    if ("clo::" in name):
        d=seen.setdefault(k,[0,0.0,r]); d[0]+=1; d[1]+=r['duration']
for k,(c,t,r) in seen.items():
    wg=r['workgroup_x']; blocks=r['grid_x']//wg*max(1,r['grid_y']//max(1,r['workgroup_y']))
    print(f"{k:46s} avg {t/c/1e3:6.2f} us  blocks {blocks:5d} x {wg:4d} thr  lds {r['lds_size']:6d}  vgpr {r['vgpr_count']:3d}")
PY
done
