R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pe -o k -f csv -- python $R/tools/prof_ekfac.py > /dev/null 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("/tmp/pe/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last rocsolver kernel marks the end of the eigh phase
last = max(i for i, r in enumerate(rows) if "rocsolver" in r["Kernel_Name"] or "stedc" in r["Kernel_Name"])
sel = rows[last + 1:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:80]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"4 correction passes: {len(sel)} kernels, {tot/1e3:.2f} ms kernel time (/4 = {tot/4e3:.2f} ms per pass)")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{c:5d} {t/4:9.1f} us/pass avg {t/c:8.1f}  {k}")
PY
