"""Secondary benchmark: KFAC factor build / inverse / matvec times (BASELINE configs C3, C4).

    python benchmarks/bench_kfac.py [lenet|resnet18] [--batch B]

Protocol of the reference's harness (`benchmark_execute.py:288-301`): min over repeats after one
warm-up, device synchronised around each phase; eval mode; Linear/Conv2d parameters only; joint
weight+bias; one MC sample."""

import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params, lenet5


def timed(fn, repeats=5):
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
    ap.add_argument("--fisher", default="mc")
    ap.add_argument("--ekfac", action="store_true", help="also time the EKFAC phases")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if args.model == "lenet":
        model, shape, B = lenet5(), (1, 32, 32), args.batch or 1024
    elif args.model == "encoder":  # 12-layer d = 768 encoder (C5 net): Linear layers with weight sharing
        from benchmarks.models import Encoder

        model, shape, B = Encoder(), (128, 768), args.batch or 8
    else:
        model, shape, B = ResNet18(), (3, 32, 32), args.batch or 512
    model = model.to(dev).eval()
    params = kfac_params(model)
    X, y = torch.rand(B, *shape, device=dev), torch.randint(0, 10, (B,), device=dev)
    data = [(X, y)]
    kw = dict(fisher_type=args.fisher, separate_weight_and_bias=False, check_deterministic=False, num_data=B)
    res = {"model": args.model, "batch": B, "D": sum(p.numel() for p in params.values())}
    res["kfac_factors_ms"], K = timed(lambda: C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, data, **kw))
    v = torch.rand(K.shape[1], device=dev)
    res["kfac_matvec_ms"], _ = timed(lambda: K @ v)
    # algorithmic work: matvec = sum_l 2 (d_out^2 d_in' + d_out d_in'^2) flop; factor
    # build = sum_l 2 B S_l (d_in'^2 + d_out^2) flop with S_l recovered from a forward pass
    blocks = [[S.shape[0] for S in blk._factors] for blk in K[1]]
    mv_flops = sum(2.0 * (b[0] ** 2 * b[1] + b[0] * b[1] ** 2) for b in blocks if len(b) == 2)
    res["kfac_matvec_tflops"] = mv_flops / (res["kfac_matvec_ms"] * 1e-3) / 1e12
    rows = {}
    def shared_positions(mod, o):  # spatial positions of a conv output, sequence positions of a linear one
        feat = o.shape[1] if isinstance(mod, nn.Conv2d) else o.shape[-1]
        return o.numel() // (o.shape[0] * feat)

    hooks = [m.register_forward_hook(lambda mod, i, o, rows=rows: rows.__setitem__(mod, shared_positions(mod, o)))
             for m in model.modules() if isinstance(m, (nn.Conv2d, nn.Linear))]
    with torch.no_grad():
        model(X[:2])
    for h in hooks:
        h.remove()
    build_flops = 0.0
    for m, S in rows.items():
        d_out = m.weight.shape[0]
        d_in = m.weight[0].numel() + (1 if m.bias is not None else 0)
        build_flops += 2.0 * B * S * (d_in**2 + d_out**2)
    res["factor_build_gflop"] = build_flops / 1e9
    res["factor_build_tflops_incl_autograd"] = build_flops / (res["kfac_factors_ms"] * 1e-3) / 1e12
    res["cholesky_inverse_ms"], Kinv = timed(lambda: K.inverse(damping=1e-3), repeats=2)
    res["inverse_matvec_ms"], _ = timed(lambda: Kinv @ v)
    if args.model == "lenet":  # BASELINE C3 also asks for the heuristic and exact damping variants
        res["inverse_heuristic_ms"], _ = timed(lambda: K.inverse(damping=1e-3, use_heuristic_damping=True), repeats=2)
        res["inverse_exact_ms"], _ = timed(lambda: K.inverse(damping=1e-3, use_exact_damping=True), repeats=2)
    # forward+backward alone (host-framework time that any backend pays)
    def fwdbwd():
        out = model(X)
        loss = nn.functional.cross_entropy(out, y)
        return torch.autograd.grad(loss, list(params.values()))
    res["gradient_and_loss_ms"], _ = timed(fwdbwd)
    if args.ekfac:
        # EKFAC = factors + eigendecompositions + eigenvalue-correction pass (reference phases of
        # benchmark_utils.py:139-143); the eigh phase is timed on the KFAC factors above
        from curvlinops_amd import linalg_native

        facs = [S for blk in K[1] for S in blk._factors]
        res["eigh_ms"], _ = timed(lambda: linalg_native.eigh_many(facs), repeats=2)
        res["ekfac_total_ms"], E = timed(lambda: C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, data, **kw),
                                         repeats=2)
        res["ekfac_matvec_ms"], _ = timed(lambda: E @ v)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
