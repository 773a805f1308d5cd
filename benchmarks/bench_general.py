"""Secondary benchmark: curvature matvecs of GENERAL nets (autograd / torch.func path on the GPU; the
native kernels only do the output-space curvature, packing and reductions) -- BASELINE configs C4
(ResNet-18 matvecs) and C5 (12-layer encoder, EF, Hutch++ with K = 32 probe blocks).

    python benchmarks/bench_general.py [resnet18|encoder] [--batch B]
"""

import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import curvlinops_amd as C
from benchmarks.models import Encoder, ResNet18, kfac_params


def timed(fn, repeats=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="resnet18")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--layers", type=int, default=12)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    res = {"model": args.model}
    if args.model == "resnet18":
        B = args.batch or 512
        model = ResNet18().to(dev).eval()
        params = kfac_params(model)
        X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
        kw = dict(check_deterministic=False, num_data=B)
        ops = {"ggn": C.GGNLinearOperator, "ef": C.EFLinearOperator, "hessian": C.HessianLinearOperator}
        res.update(batch=B, D=sum(p.numel() for p in params.values()))
        for name, cls in ops.items():
            op = cls(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
            v = torch.rand(op.shape[1], device=dev)
            res[f"{name}_matvec_ms"], _ = timed(lambda: op @ v)
    else:
        B = args.batch or 8
        model = Encoder(layers=args.layers).to(dev).eval()
        params = dict(model.named_parameters())
        X, y = torch.rand(B, 128, 768, device=dev), torch.randint(0, 10, (B,), device=dev)
        op = C.EFLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], check_deterministic=False, num_data=B)
        D = op.shape[1]
        res.update(batch=B, D=D, layers=args.layers)
        v = torch.rand(D, device=dev)
        res["ef_matvec_ms"], _ = timed(lambda: op @ v)
        print(json.dumps(res), file=sys.stderr, flush=True)
        V = torch.rand(D, 32, device=dev)
        res["ef_matmat_k32_ms"], _ = timed(lambda: op @ V, repeats=1)
        del V
        print(json.dumps(res), file=sys.stderr, flush=True)
        best = float("inf")
        for _ in range(2):  # min of two: the first call also pays for 10 GB-sized first allocations
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr = C.hutchpp_trace(op, num_matvecs=96)
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0))
        res["hutchpp_96_ms"] = best
        res["hutchpp_trace"] = float(tr)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
