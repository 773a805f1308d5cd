# kernel statistics of the tridiagonalisation / eigensolver at the orders given (default 4609)
cd /tmp && export TMPDIR=/tmp
N="${@:-4609}"
rm -rf /tmp/prof_sytrd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sytrd -o sy -- python $GRAFT_REPO_ROOT/tools/probe_sytrd.py $N > /tmp/prof_sytrd.log 2>&1
grep "^n=" /tmp/prof_sytrd.log
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_sytrd/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):7d} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
