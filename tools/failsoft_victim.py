"""The victim of the fail-soft test: products on the persistent MLP kernel while another process holds most CUs.
Prints one JSON line: what each step did (no exception may escape, the context must stay usable)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip

lib = _hip.load()
dims, acts = [64, 256, 512, 10], [1, 1, 0]
torch.manual_seed(0)
W = [torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5 for i in range(3)]
b = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(3)]
VW = [torch.rand_like(w) for w in W]
Vb = [torch.rand_like(x) for x in b]
X = torch.rand(8, dims[0], device="cuda")
def product(flags):
    plan = product.plans.setdefault(flags, _hip.MLPPlan(dims, acts))
    plan.flags = flags
    OW, Ob = [torch.empty_like(w) for w in W], [torch.empty_like(x) for x in b]
    plan.ggn_matvec(W, b, VW, Vb, OW, Ob, X, 0, 2.0 / 80, 1.0, 0.0)
    torch.cuda.synchronize()
    return torch.cat([t.flatten() for t in OW + Ob])
product.plans = {}
out = {}
ref = product(_hip.MLP_NO_PERSISTENT)                 # launch chain
ok = product(_hip.MLP_DEFAULT)                        # persistent kernel, GPU free
out["persistent_equals_chain_when_free"] = float((ok - ref).abs().max() / ref.abs().max()) < 1e-5
out["status_before"] = _hip.persistent_status()
assert lib.clo_test_set_spin_limit(1 << 15) == 0      # ~0.1 s instead of seconds
print("victim ready", flush=True)
sys.stdin.readline()                                   # the test starts the hog and tells us when it runs
t0 = time.time()
try:
    bad = product(_hip.MLP_DEFAULT)                    # cannot become co-resident: must END (garbage), not trap
    out["timed_out_launch_returned"] = True
    out["timed_out_result_has_nan"] = bool(torch.isnan(bad).any())   # a timed-out product is loud, not plausible
    out["timed_out_seconds"] = time.time() - t0
except Exception as e:  # noqa: BLE001
    out["timed_out_launch_returned"] = False
    out["error_at_launch"] = repr(e)
try:
    product(_hip.MLP_DEFAULT)
    out["reported"] = False
except RuntimeError as e:
    out["reported"] = "timed out" in str(e)
    out["message"] = str(e)[:160]
out["status_after"] = _hip.persistent_status()
again = product(_hip.MLP_DEFAULT)                      # now served by the launch chain
out["next_product_equals_chain"] = float((again - ref).abs().max() / ref.abs().max()) < 1e-6
out["context_alive"] = bool(torch.isfinite(torch.ones(4, device="cuda").sum()))
print(json.dumps(out), flush=True)
