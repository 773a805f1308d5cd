"""Do two hardware queues that sit on the same dispatch pipe slow each other down?  A dispatch-bound kernel (40 k workgroups
of ~1 us) runs on the null stream and, at the same time, on each of 12 freshly created streams in turn."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1] if len(sys.argv) > 1 else "16"
import torch
from curvlinops_amd import _hip
lib = _hip.load()
torch.zeros(1, device="cuda")
null = torch.cuda.current_stream()
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cands = [torch.cuda.Stream(priority=prio) for _ in range(12)]
def run(st, n=40000):
    assert lib.clo_test_occupy(n, 0, 100, st.cuda_stream) == 0
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
for s in cands:   # first use (queue creation) outside the timing
    run(s, 10)
torch.cuda.synchronize()
single = min(timed(lambda: run(null)) for _ in range(3))
print(f"queues={os.environ['GPU_MAX_HW_QUEUES']} prio={prio}: null stream alone {single:.2f} ms")
for i, s in enumerate(cands):
    def pair():
        run(null); run(s)
    t = min(timed(pair) for _ in range(3))
    print(f"  null + stream {i:2d}: {t:.2f} ms ({t / single:.2f} x alone)")
