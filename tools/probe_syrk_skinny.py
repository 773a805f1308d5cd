import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
for rows, d, ones in [(8_028_160, 6, False), (802_816, 6, False), (802_816, 25, True), (1_024_000, 16, False), (102_400, 150, True),
                      (524_288, 64, False), (524_288, 576, True), (85_000_000, 32, False)]:
    X = torch.randn(rows, d, device="cuda")
    dd = d + int(ones)
    C = torch.zeros(dd, dd, device="cuda")
    for _ in range(2): _hip.syrk_accum(C, X, alpha=1.0, beta=0.0, ones_col=ones)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): _hip.syrk_accum(C, X, alpha=1.0, beta=0.0, ones_col=ones)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print(f"syrk rows={rows:9d} d={dd:4d}: {t*1e3:8.3f} ms  read {rows*d*4/t/1e12:.2f} TB/s  {2*rows*dd*dd/t/1e12:.1f} TF")
    del X
