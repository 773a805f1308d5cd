"""Two 4609-order factors at once: both on rocSOLVER, both on the own reduction, one each (two host threads / streams)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import linalg_native as L
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4609
mats = []
for i in range(2):
    X = torch.randn(2048, n, device=dev)
    mats.append(X.T @ X / 2048)
def run(fns):
    streams = [torch.cuda.Stream() for _ in fns]
    def work(i):
        with torch.cuda.stream(streams[i]):
            fns[i](mats[i])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
    torch.cuda.synchronize(); t = time.perf_counter()
    for th in ts: th.start()
    for th in ts: th.join()
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
roc, own = L._torch_eigh_scaled, L.eigh_sytrd
for name, fns in (("rocSOLVER + rocSOLVER", [roc, roc]), ("own + own", [own, own]), ("rocSOLVER + own", [roc, own])):
    run(fns)
    print(f"n={n}: {name}: {min(run(fns) for _ in range(3)):.1f} ms", flush=True)
