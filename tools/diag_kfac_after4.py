import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from curvlinops_amd import computers
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
bench.secondary_configs(dev)
out = bench.kfac_leg(dev, 1, 0)
c = [v for v in computers._CAPTURED.values() if hasattr(v, "replay_ms")]
print(f"skip={os.environ.get('CLO_BENCH_SKIP','')}: build {out['ms_per_batch']:.2f} ms, captured builds {len(computers._CAPTURED)}, tuning " + "; ".join(f"{v.replay_ms:.2f} (one branch {v.serial_ms:.2f}, {v.tries} tries)" for v in c), flush=True)
