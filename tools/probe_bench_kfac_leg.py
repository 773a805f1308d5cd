"""bench.py's kfac leg on its own (same process setup as bench.py: GPU_MAX_HW_QUEUES etc.), optionally after the legs that
precede it in the bench process (argv[1] = "after": C3 / C4-EKFAC / C5 secondary configs first)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # sets GPU_MAX_HW_QUEUES before torch touches the device
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if len(sys.argv) > 1 and sys.argv[1] == "after":
    sec = bench.secondary_configs(dev)
    print("secondary:", {k: {kk: vv for kk, vv in v.items() if kk.endswith("_ms")} for k, v in sec.items() if isinstance(v, dict)}, flush=True)
out = bench.kfac_leg(dev, 1, 0)
print(json.dumps({k: out[k] for k in ("ms_per_batch", "gradient_and_loss_ms", "kfac_matvec_ms", "cholesky_inverse_ms_first_call",
                                      "cholesky_inverse_ms_second_call", "cholesky_inverse_ms_mean_of_4")}), flush=True)
