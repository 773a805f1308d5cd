"""Round-4 crash hunt, suspect 2: the helper-stream pool of the damped Cholesky inverses.  REPS batches of mixed factor
sizes go through linalg_native.concurrent_inverses (worker threads with their own streams, each big factor a pipeline
over a helper stream) while a third party keeps the default stream busy; every result must equal the serial one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import linalg_native as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
sizes = [64, 65, 128, 147, 577, 576, 1153, 1152, 2305, 2304, 4609, 256, 513, 10]
mats = []
for n in sizes:
    X = torch.randn(2 * n, n, device=dev, generator=g)
    mats.append((X.T @ X) / (2 * n))
serial = [L.damped_cholesky_inverse(A, 1e-3) for A in mats]
torch.cuda.synchronize()
busy = torch.randn(4096, 4096, device=dev)
t0 = time.perf_counter()
for rep in range(reps):
    order = torch.randperm(len(mats)).tolist()
    busy2 = busy @ busy                      # default-stream work beside the batch
    with L.concurrent_inverses(num_streams=2 + rep % 3):
        outs = [L.damped_cholesky_inverse(mats[i], 1e-3) for i in order]
    torch.cuda.synchronize()
    for i, o in zip(order, outs):
        if not torch.equal(o, serial[i]):
            err = float((o - serial[i]).abs().max() / serial[i].abs().max())
            raise SystemExit(f"rep {rep}: factor of order {sizes[i]} differs from the serial result (rel {err:.2e})")
print(f"{reps} batches of {len(mats)} factors: every inverse bit-equal to the serial call ({time.perf_counter() - t0:.1f} s)")
