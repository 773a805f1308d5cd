"""Why is bench.py's kfac leg slower after the secondary configs?  kfac_leg alone, after secondary_configs, after dropping
the captured builds, after emptying the allocator cache."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from curvlinops_amd import computers
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
def leg(tag):
    out = bench.kfac_leg(dev, 1, 0)
    print(f"{tag}: build {out['ms_per_batch']:.2f} ms, cholesky second call {out['cholesky_inverse_ms_second_call']:.1f} ms, "
          f"captured builds {len(computers._CAPTURED)}, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
leg("alone")
sec = bench.secondary_configs(dev)
leg("after secondary")
leg("again")
computers.reset_captured_builds(); gc.collect()
leg("after reset_captured_builds")
torch.cuda.empty_cache()
leg("after empty_cache")
