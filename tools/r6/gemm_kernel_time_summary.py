"""usage: gemm_kernel_time_summary.py results.db probe_stdout.txt"""
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end, grid_x from kernels order by start"))
shapes = [tuple(map(int, m.groups())) for m in re.finditer(r"SHAPE (\d+) (\d+) (\d+)", open(sys.argv[2]).read())]
host = [tuple(map(float, m.groups())) for m in re.finditer(r"clo ([\d.]+) torch ([\d.]+)", open(sys.argv[2]).read())]
fills = [i for i, r in enumerate(rows) if "FillFunctor<float>" in r[0]]
gx = max(set(rows[i][3] for i in fills), key=lambda g: sum(1 for i in fills if rows[i][3] == g))
marks = [i for i in fills if rows[i][3] == gx][-(2 * len(shapes) + 1):]
print("# kernel time per call (sum of the durations of the kernels of one call, rocprofv3 --kernel-trace) and the interval from the first")
print("# kernel's start to the last kernel's end / 10 (includes the gaps between launches); host = Python time to issue one call")
print("# M N K | clo kernels us (names) interval us host us | hipBLASLt kernels us interval us host us | clo / hipBLASLt (kernels)")
for si, (M, N, K) in enumerate(shapes):
    res = []
    for leg in range(2):
        a, b = marks[2 * si + leg], marks[2 * si + leg + 1]
        sel = [r for r in rows[a + 1:b] if "FillFunctor<float>" not in r[0]]
        ksum = sum(r[2] - r[1] for r in sel) / 1e3 / 10
        wall = (sel[-1][2] - sel[0][1]) / 1e3 / 10
        names = sorted(set(re.sub(r"^void ", "", r[0]).split("(")[0][:60] for r in sel))
        res.append((ksum, wall, len(sel) // 10, names))
    fl = 2.0 * M * N * K
    print(f"{M:5d} {N:5d} {K:5d} | clo {res[0][0]:7.1f} us {fl / res[0][0] / 1e6:6.1f} TF ({res[0][2]} launches: {'; '.join(res[0][3])}) interval {res[0][1]:7.1f} host {host[si][0]:6.1f}"
          f" | hipBLASLt {res[1][0]:7.1f} us {fl / res[1][0] / 1e6:6.1f} TF ({res[1][2]} launches) interval {res[1][1]:7.1f} host {host[si][1]:6.1f} | {res[0][0] / res[1][0]:4.2f}")
