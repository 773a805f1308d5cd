# SQ counters of the GEMM kernel of a 1024^3 fp32 product (library named by CLO_HIP_LIB or the default): where do the waves wait?
export TMPDIR=/tmp; R=$PWD; cd /tmp
cat > /tmp/g1024.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from curvlinops_amd import _hip
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
for _ in range(5): _hip.gemm(A, B, out=out)
torch.cuda.synchronize()
PY
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  rm -rf /tmp/pg; rocprofv3 --pmc $c --kernel-trace -d /tmp/pg -o g -- python /tmp/g1024.py ${SHAPE:-1024 1024 1024} > /dev/null 2>&1
  python - <<PY
import sqlite3
con = sqlite3.connect("/tmp/pg/g_results.db")
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if "pmc" in x.lower() or "counter" in x.lower()]
try:
    rows = list(con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
except Exception as e:
    rows = []; print("tables:", t, e)
for k, c, v, n in rows:
    if "gemm" in k: print(f"{c:28s} {v / max(n,1):14.0f}  per launch x{n}  {k[:70]}")
PY
done
