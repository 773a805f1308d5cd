# kernel composition of the EKFAC eigenvalue-correction sweep (ResNet-18, 512 rows): rocprofv3 --kernel-trace
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe
rocprofv3 --kernel-trace -d /tmp/pe -o k -f csv -- python $R/tools/prof_ekfac.py 2>&1 | grep "correction pass"
python - <<'PY'
import csv, collections, glob
f = glob.glob("/tmp/pe/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = ("sytrd", "tql2", "dc_", "ormtr", "larft")
last = max(i for i, r in enumerate(rows) if any(m in r["Kernel_Name"] for m in marks))
sel = rows[last + 1:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
print(f"4 correction passes: {len(sel)} kernels, {tot/1e3:.2f} ms kernel time (/4 = {tot/4e3:.2f} ms per pass), span {span:.1f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{c:5d} {t/4:9.1f} us/pass avg {t/c:8.1f}  {k}")
PY
