import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
rows = list(con.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels where name like '%gemm_f32_kernel<false>%'")) if 'grid_y' in cols else list(con.execute("select name, start, end, grid_x, 0, 0 from kernels where name like '%gemm_f32_kernel<false>%'"))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gx, gy, wx in rows:
    agg[(gx, gy)][0] += 1; agg[(gx, gy)][1] += (e - s) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"grid {k}: {c} calls, {t / c:.1f} us avg, {t:.0f} us total")
# what runs right before each of the biggest ones
allk = list(con.execute("select name, start, end from kernels order by start"))
idx = {s: i for i, (n, s, e) in enumerate(allk)}
big = sorted(rows, key=lambda r: -(r[2] - r[1]))[:3]
for r in big:
    i = idx[r[1]]
    print("around the longest:", [allk[j][0][:50] for j in range(max(0, i - 3), min(len(allk), i + 3))])
