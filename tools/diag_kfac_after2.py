"""bench.py's kfac leg captured AFTER the secondary configs: is the slow replay tied to the capture (re-capture changes
it?) or to the process state?"""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from curvlinops_amd import computers
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
def leg(tag):
    out = bench.kfac_leg(dev, 1, 0)
    print(f"{tag}: build {out['ms_per_batch']:.2f} ms, captured builds {len(computers._CAPTURED)}", flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which == "all":
    bench.secondary_configs(dev)
elif which == "streams":
    ss = [torch.cuda.Stream() for _ in range(int(sys.argv[2]))]
    for s in ss:
        with torch.cuda.stream(s): torch.zeros(8, device=dev).add_(1)
    torch.cuda.synchronize()
leg("first capture after " + which)
for i in range(3):
    computers.reset_captured_builds(); gc.collect()
    leg(f"re-capture {i}")
computers._CAPTURE = False
leg("eager")
