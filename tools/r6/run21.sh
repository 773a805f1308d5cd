R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r21
cd /tmp && export TMPDIR=/tmp
for n in 33 64; do
rm -rf /tmp/pr$n
rocprofv3 --kernel-trace --stats -d /tmp/pr$n -o k -- python $R/tools/probe_c2.py $n > /tmp/probe_$n.log 2>&1
grep "N=" /tmp/probe_$n.log
python $R/tools/prof_summary.py /tmp/pr$n/k_results.db $R/gpurun_out/r21/c2_n${n}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_c2.py $n"
cut -c1-150 $R/gpurun_out/r21/c2_n${n}_kernel_stats.txt | head -11
python - <<PY
import sqlite3
con=sqlite3.connect("/tmp/pr$n/k_results.db")
seen=set()
cur=con.execute("select * from kernels limit 1"); cols=[d[0] for d in cur.description]; print(cols)
for row in con.execute("select * from kernels"):
    r=dict(zip(cols,row)); name=r['name']
    k=name.split('(')[0][-40:]
    if ('mid_' in name or 'head_' in name) and k not in seen:
        seen.add(k); print(k, {c:r[c] for c in cols if any(t in c.lower() for t in ('grid','workgroup','lds','vgpr','scratch'))})
PY
done
