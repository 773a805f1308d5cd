# timeline of the (2305 x 4) unit inside the back-to-back sequence of tools/probe_chol_sequence.py
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pcs
rocprofv3 --kernel-trace -d /tmp/pcs -o k -f csv -- python $R/tools/probe_chol_sequence.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pcs/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
init = [i for i, r in enumerate(rows) if "chol_init" in r["Kernel_Name"]]
# the last four chol_init launches = the last per-unit pass; the second of them starts the 2305 x 4 unit
a, b = init[-3], init[-2]
sel = rows[a:b]
t0 = int(sel[0]["Start_Timestamp"])
print(f"unit 2305 x 4: {len(sel)} kernels, span {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us")
for r in sel:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("clo::", "")[:40]
    print(f"{(s_ - t0) / 1e3:8.1f} {(e_ - t0) / 1e3:8.1f} q{r.get('Queue_Id', '?')} {nm} g{r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
PY
