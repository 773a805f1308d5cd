"""Kernel timeline of the LAST K-column product in a rocprofv3 --kernel-trace results .db of tools/probe_cols.py: start
(us since the product's first kernel), duration, queue, name -- shows which launches overlap."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
rows = list(con.execute(f"select name, start, end, {qcol or 0} from kernels order by start"))
# the last product: from the last kfwd_stream_kernel<false (layer 1) backwards to the fwd kernels before it
idx = [i for i, r in enumerate(rows) if "kfwd_stream_kernel<false" in r[0]]
i0 = idx[-1]
while i0 > 0 and ("fwd_mfma" in rows[i0 - 1][0] or "fwd_finish" in rows[i0 - 1][0]): i0 -= 1
t0 = rows[i0][1]
for n, s, e, q in rows[i0:]:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  q{q}  {n[:90]}")
