cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_km
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -o km -- python $GRAFT_REPO_ROOT/tools/probe_kfac_matvec_prof.py 2>&1 | grep "kfac matvec"
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_km/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows=[r for r in rows if not any(r['Name'].startswith(x) or x in r['Name'][:40] for x in ('naive_conv','Cijk','miopen','ck::','_ZN2ck','Im2d','Col2Im','void ck','SubTensor','batched_transpose','MIOpen','gridwise','igemm','transpose_'))]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:84]:84s} calls {int(r['Calls']):6d} total {float(r['TotalDurationNs'])/1e6:8.2f} ms ({100*float(r['TotalDurationNs'])/tot:4.1f}%) avg {float(r['AverageNs'])/1e3:7.2f} us")
PY
