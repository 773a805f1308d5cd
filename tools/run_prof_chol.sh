# kernel timeline of ONE damped Cholesky inverse (n = $1, default 4608): rocprofv3 --kernel-trace, last call only
R=$PWD; N=${1:-4608}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc
rocprofv3 --kernel-trace -d /tmp/pc -o k -f csv -- python $R/tools/probe_chol_one.py $N > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pc/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
init = [i for i, r in enumerate(rows) if "chol_init" in r["Kernel_Name"]]
sel = rows[init[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in sel)
print(f"last call: {len(sel)} kernels, span {(end - t0) / 1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:5d} {t:9.1f} us avg {t / c:7.1f}  {k}")
nodes = [r for r in sel if "potrf_node128" in r["Kernel_Name"] or "potrf_diag" in r["Kernel_Name"]]
print("node-to-node start intervals (us):", " ".join(f"{(int(b['Start_Timestamp']) - int(a['Start_Timestamp'])) / 1e3:.0f}" for a, b in zip(nodes, nodes[1:])))
print(f"first node start {(int(nodes[0]['Start_Timestamp']) - t0) / 1e3:.1f} us, last node end {(int(nodes[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
print("all kernels of the call (start, end in us from the first; queue; name; grid):")
for r in sel:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("clo::", "")[:44]
    print(f"ALL {(s_ - t0) / 1e3:8.1f} {(e_ - t0) / 1e3:8.1f} q{r.get('Queue_Id', '?')} {nm} g{r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
lastn = int(nodes[-1]["Start_Timestamp"])
print("tail (from the last node on):")
for r in sel:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e_ > lastn - 200000:
        print(f"   {(s_ - lastn) / 1e3:8.1f} .. {(e_ - lastn) / 1e3:8.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
# what runs between two nodes on the critical path (kernels starting after node k ends and before node k+1 starts)
k = len(nodes) // 3
a, b = int(nodes[k]["End_Timestamp"]), int(nodes[k + 1]["Start_Timestamp"])
print(f"between node {k} and {k + 1}:")
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > a - 60000 and s < b + 1000:
        print(f"   {(s - a) / 1e3:8.1f} .. {(e - a) / 1e3:8.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]}")
PY
