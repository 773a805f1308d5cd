"""Parity on the network FAMILIES of BASELINE configs C4 (ResNet-18: BatchNorm in eval mode, residual
blocks, strided 1x1 down-sampling) and C5 (pre-LN transformer encoder) against golden vectors the
reference produced on toy instances of the very same module classes (`oracle/make_golden_nets.py` ->
`tests/golden/nets.npz`), on CPU in float64 and -- marked `gpu` -- on the MI355X in float32 through the
HIP kernels; plus size-independent properties at the FULL BASELINE sizes on the GPU.

Tolerances: products / factors 1e-4, damped inverses 1e-3 in fp32; 1e-7 in float64.
"""

import numpy as np
import pytest
import torch
from torch import nn

import curvlinops_amd as C
from benchmarks.models import Encoder, ResNet18, ResNetToy, encoder_toy, kfac_params, lenet5
from conftest import load_golden
from helpers import rel_err

F32, F64 = torch.float32, torch.float64


def _load(model: nn.Module, rec: dict, dtype, device) -> nn.Module:
    sd = {k[len("state:"):]: torch.as_tensor(v) for k, v in rec.items() if k.startswith("state:")}
    model = model.to(torch.float64)  # load at full precision, then cast
    model.load_state_dict(sd)
    return model.to(device=device, dtype=dtype).eval()


def _data(rec: dict, dtype, device):
    return [(torch.as_tensor(rec[f"X{i}"], dtype=dtype, device=device), torch.as_tensor(rec[f"y{i}"], device=device))
            for i in range(int(rec["num_batches"]))]


def _t(x, dtype, device):
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(device)


# ------------------------------------------------------------------------------------------ bodies
def check_resnet_toy(dtype, device, tol, tol_inv, tol_ekfac_inv):
    rec = load_golden("nets")["resnet_toy"]
    model = _load(ResNetToy(), rec, dtype, device)
    params = kfac_params(model)
    data = _data(rec, dtype, device)
    V = _t(rec["V"], dtype, device)
    lf = nn.CrossEntropyLoss()
    for fisher in ("empirical", "type-2"):
        for sep in (True, False):
            tag = f"{fisher}|{'sep' if sep else 'joint'}"
            K = C.KFACLinearOperator(model, lf, params, data, fisher_type=fisher, separate_weight_and_bias=sep,
                                     check_deterministic=False)
            for b, block in enumerate(K[1]):
                for f, fac in enumerate(block):
                    assert rel_err(fac, rec[f"kfac|{tag}/block{b}_factor{f}"]) < tol, (tag, b, f)
            assert rel_err(K @ V, rec[f"kfac|{tag}/KV"]) < tol, tag
            assert rel_err(K.trace(), rec[f"kfac|{tag}/trace"]) < tol, tag
            assert rel_err(K.inverse(damping=1e-2) @ V, rec[f"kfac|{tag}/inv_plain"]) < tol_inv, tag
            got = K.inverse(damping=1e-2, use_heuristic_damping=True, min_damping=1e-4) @ V
            assert rel_err(got, rec[f"kfac|{tag}/inv_heur"]) < tol_inv, tag
            got = K.inverse(damping=1e-2, use_exact_damping=True) @ V
            assert rel_err(got, rec[f"kfac|{tag}/inv_exact"]) < tol_inv, tag
            E = C.EKFACLinearOperator(model, lf, params, data, fisher_type=fisher, separate_weight_and_bias=sep,
                                      check_deterministic=False)
            assert rel_err(E.trace(), rec[f"ekfac|{tag}/trace"]) < tol_inv, tag
            assert rel_err(E @ V, rec[f"ekfac|{tag}/EV"]) < tol_inv, tag
            assert rel_err(E.inverse(damping=1e-2) @ V, rec[f"ekfac|{tag}/invEV"]) < tol_ekfac_inv, tag
    G = C.GGNLinearOperator(model, lf, params, data, check_deterministic=False)
    assert rel_err(G @ V, rec["ggn/GV"]) < tol
    Fm = C.EFLinearOperator(model, lf, params, data, check_deterministic=False)
    assert rel_err(Fm @ V, rec["ef/FV"]) < tol
    assert all(p.grad is None for p in model.parameters())


def check_encoder_toy(dtype, device, tol, tol_inv, tol_ekfac_inv):
    rec = load_golden("nets")["encoder_toy"]
    model = _load(encoder_toy(), rec, dtype, device)
    params = dict(model.named_parameters())
    data = _data(rec, dtype, device)
    V = _t(rec["V"], dtype, device)
    lf = nn.CrossEntropyLoss()
    EF = C.EFLinearOperator(model, lf, params, data, check_deterministic=False)
    assert rel_err(EF @ V, rec["ef/FV"]) < tol
    assert rel_err(EF @ V[:, 0].contiguous(), rec["ef/Fv"]) < tol
    GG = C.GGNLinearOperator(model, lf, params, data, check_deterministic=False)
    assert rel_err(GG @ V, rec["ggn/GV"]) < tol
    pool = _t(rec["ef/pool"], dtype, device)
    got = C.hutchpp_trace(EF, 12, "rademacher", probes=(pool[:, :4].contiguous(), pool[:, 4:8].contiguous()))
    assert rel_err(got, rec["ef/hutchpp"]) < tol_inv
    lin = kfac_params(model)
    Vl = _t(rec["Vlin"], dtype, device)
    for fisher in ("empirical", "type-2"):
        tag = f"{fisher}|joint"
        K = C.KFACLinearOperator(model, lf, lin, data, fisher_type=fisher, separate_weight_and_bias=False,
                                 check_deterministic=False)
        for b, block in enumerate(K[1]):
            for f, fac in enumerate(block):
                assert rel_err(fac, rec[f"kfac|{tag}/block{b}_factor{f}"]) < tol, (tag, b, f)
        assert rel_err(K @ Vl, rec[f"kfac|{tag}/KV"]) < tol, tag
        assert rel_err(K.inverse(damping=1e-2) @ Vl, rec[f"kfac|{tag}/inv_plain"]) < tol_inv, tag
        E = C.EKFACLinearOperator(model, lf, lin, data, fisher_type=fisher, separate_weight_and_bias=False,
                                  check_deterministic=False)
        assert rel_err(E @ Vl, rec[f"ekfac|{tag}/EV"]) < tol_inv, tag
        assert rel_err(E.inverse(damping=1e-2) @ Vl, rec[f"ekfac|{tag}/invEV"]) < tol_ekfac_inv, tag


# ------------------------------------------------------------------------------------------ CPU, float64
def check_kfac_mc_replayed(name, dtype, device, tol, tol_inv, monkeypatch):
    """``fisher_type="mc"`` with the reference's SAMPLED backprop vectors replayed (tests/golden/kfac_mc.npz,
    oracle/make_golden_nets.py::gen_kfac_mc): the MC code path itself -- M vectors per datum, the 1/sqrt(M) scale, the
    mean-reduction correction -- against the reference's factors, product and damped inverse, not "in expectation"."""
    from curvlinops_amd import computers

    rec = load_golden("kfac_mc")[name]
    model = _load(lenet5() if name == "lenet5" else ResNetToy(), rec, dtype, device)
    params = dict(model.named_parameters()) if name == "lenet5" else kfac_params(model)
    data = _data(rec, dtype, device)
    V = _t(rec["V"], dtype, device)
    queue = []

    def replaying_vmap(fn, **kwargs):   # stands in for vmap(make_grad_output_fn(...)): [M, B, C] per mini-batch
        def replay(output, y, generator):
            g = queue.pop(0)
            assert g.shape[1:] == output.shape
            return g.clone()
        return replay

    monkeypatch.setattr(computers, "vmap", replaying_vmap)
    for M in (1, 3):
        for sep in ((False,) if name == "lenet5" else (True, False)):
            tag = f"mc{M}|{'sep' if sep else 'joint'}"
            queue[:] = [_t(rec[f"{tag}/grad_outputs{i}"], dtype, device) for i in range(len(data))]
            K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, data, fisher_type="mc", mc_samples=M,
                                     separate_weight_and_bias=sep, check_deterministic=False)
            assert not queue, "the sampled vectors were not consumed"
            for b, block in enumerate(K[1]):
                for f, fac in enumerate(block):
                    assert rel_err(fac, rec[f"{tag}/block{b}_factor{f}"]) < tol, (tag, b, f)
            assert rel_err(K @ V, rec[f"{tag}/KV"]) < tol, tag
            assert rel_err(K.inverse(damping=1e-2) @ V, rec[f"{tag}/inv_plain"]) < tol_inv, tag


@pytest.mark.parametrize("name", ["lenet5", "resnet_toy"])
def test_kfac_mc_replayed_vectors_cpu(name, monkeypatch):
    check_kfac_mc_replayed(name, F64, torch.device("cpu"), 2e-6, 2e-6, monkeypatch)   # (large arrays stored in float32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lenet5", "resnet_toy"])
def test_kfac_mc_replayed_vectors_gpu(name, monkeypatch, dev):
    check_kfac_mc_replayed(name, F32, dev, 1e-4, 1e-3, monkeypatch)


def test_resnet_toy_cpu():
    check_resnet_toy(F64, torch.device("cpu"), 1e-7, 1e-7, 1e-6)


def test_encoder_toy_cpu():
    check_encoder_toy(F64, torch.device("cpu"), 1e-7, 1e-7, 1e-6)


# ------------------------------------------------------------------------------------------ GPU, float32
@pytest.fixture(scope="module")
def dev():
    from curvlinops_amd import _hip

    _hip.load()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_resnet_toy_gpu(dev):
    check_resnet_toy(F32, dev, 1e-4, 1e-3, 1e-3)


@pytest.mark.gpu
def test_encoder_toy_gpu(dev):
    check_encoder_toy(F32, dev, 1e-4, 1e-3, 1e-3)


@pytest.mark.gpu
def test_resnet18_full_size_properties(dev):
    """BASELINE C4: ResNet-18, 512 rows per GPU, joint W+b.  Size-independent properties: factors are
    symmetric PSD; the factors of two half shards (global `num_data`) sum to the full-batch factors --
    the identity the multi-GPU all-reduce relies on; the inverse of the damped operator inverts it; the
    EKFAC trace is the sum of the corrected eigenvalues and equals the trace of the exact EF blocks."""
    torch.manual_seed(0)
    model = ResNet18().to(dev).eval()
    params = kfac_params(model)
    B = 512
    X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
    kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
    lf = nn.CrossEntropyLoss()
    K = C.KFACLinearOperator(model, lf, params, [(X, y)], **kw)
    facs = [S for blk in K[1] for S in blk]
    assert len(facs) == 42 and max(S.shape[0] for S in facs) == 4608
    for S in facs:
        assert torch.equal(S, S.T)
        assert torch.isfinite(S).all() and float(S.diagonal().min()) >= 0.0
    small = [S for S in facs if S.shape[0] <= 600]
    for S in small:
        assert float(torch.linalg.eigvalsh(S.double()).min()) > -1e-5 * float(S.diagonal().max())
    halves = [C.KFACLinearOperator(model, lf, params, [(X[i:i + B // 2], y[i:i + B // 2])], num_data=B, **kw)
              for i in (0, B // 2)]
    for i, (S, S0, S1) in enumerate(zip(facs, *[[T for blk in H[1] for T in blk] for H in halves])):
        # A-type factors are plain sums over rows; G-type factors carry the (B T)^2 / (T N) correction,
        # which also makes the shard factors add up (kfac_math.py:172-203)
        # (tolerance: the shards run PyTorch's fp32 convolutions / BatchNorm at another batch size, and the
        # gradient covariances of the early layers see that noise through 18 layers of backprop)
        # blocks are [G_l, A_l].  Measured (tools/probe_shard_sum.py, probe_g_factor_source.py): the kernels are
        # exact to 1e-7 on the gradients they are given and MIOpen's convolutions to 1e-6; what differs between
        # runs at another batch size (another convolution algorithm, another rounding) is the SIGN of a
        # pre-activation that is zero to rounding -- one ReLU flip changes one sample's gradient below that
        # layer, i.e. the gradient covariances by ~1/B = 2e-3
        assert rel_err(S0 + S1, S.double().cpu().numpy()) < (1e-2 if i % 2 == 0 else 2e-3), i
    v = torch.rand(K.shape[1], device=dev)
    Kd = K.inverse(damping=1e-2)
    w = Kd @ v
    from curvlinops_amd.diag import DiagonalLinearOperator  # noqa: F401  (damped product by hand below)
    P, Kc, PT = K
    # (K + damping) applied block-wise to the canonical vector: A (x) G + 1e-2 I on every factor pair is
    # NOT what `inverse` damps (it damps each factor), so check the factor-wise identity instead
    for blk, blk_inv in zip(Kc, Kd[1]):
        for S, Sinv in zip(blk, blk_inv):
            n = S.shape[0]
            if n > 1200:
                continue
            I = torch.eye(n, device=dev)
            assert rel_err((S + 1e-2 * I) @ Sinv, I.cpu().numpy()) < 2e-3
    assert torch.isfinite(w).all()
    E = C.EKFACLinearOperator(model, lf, params, [(X[:64], y[:64])], **kw)
    lam_sum = sum(float(blk.eigenvalues.double().sum()) for blk in E[1])
    assert abs(float(E.trace()) - lam_sum) <= 1e-4 * abs(lam_sum)
    # corrected eigenvalues are second moments in the eigenbasis: non-negative, and their sum is the trace
    # of the exact per-layer EF blocks = sum_n ||grad_n of the layer||^2 / N
    grads_sq = 0.0
    for n in range(0, 64, 16):
        for i in range(n, n + 16):
            out = model(X[i:i + 1])
            g = torch.autograd.grad(nn.functional.cross_entropy(out, y[i:i + 1]), list(params.values()))
            grads_sq += sum(float(t.double().square().sum()) for t in g)
    assert abs(lam_sum - grads_sq / 64) <= 2e-3 * abs(grads_sq / 64)


@pytest.mark.gpu
def test_encoder_full_size_properties(dev):
    """BASELINE C5: 12-layer d = 768 encoder (D = 85 M), 8 sequences of 128, `EFLinearOperator`:
    symmetry `<u, F v> = <v, F u>`, PSD, linearity over a K = 4 block, `F @ v` = `sum_n g_n <g_n, v> / N`
    from per-sample gradients, and the shard-sum identity behind the data-parallel all-reduce."""
    torch.manual_seed(0)
    model = Encoder().to(dev).eval()
    params = dict(model.named_parameters())
    N = 8
    X, y = torch.rand(N, 128, 768, device=dev), torch.randint(0, 10, (N,), device=dev)
    lf = nn.CrossEntropyLoss()
    EF = C.EFLinearOperator(model, lf, params, [(X, y)], check_deterministic=False)
    D = EF.shape[1]
    assert D == 85_063_690
    u, v = torch.rand(D, device=dev), torch.rand(D, device=dev)
    Fu, Fv = EF @ u, EF @ v
    a, b = float(torch.dot(u.double(), Fv.double())), float(torch.dot(v.double(), Fu.double()))
    assert abs(a - b) <= 1e-4 * abs(a) and a >= 0.0
    M = torch.stack([u, v, u - v, 2 * u + v], dim=1)
    FM = EF @ M
    assert rel_err(FM[:, 2], (Fu - Fv).double().cpu().numpy()) < 2e-4
    assert rel_err(FM[:, 3], (2 * Fu + Fv).double().cpu().numpy()) < 2e-4
    # per-sample gradients: F v = (1/N) sum_n g_n <g_n, v> with g_n the gradient of the n-th loss term
    plist = list(params.values())
    ref = torch.zeros(D, device=dev, dtype=torch.float64)
    for n in range(N):
        out = model(X[n:n + 1])
        g = torch.cat([t.reshape(-1) for t in torch.autograd.grad(nn.functional.cross_entropy(out, y[n:n + 1]), plist)])
        ref += g.double() * torch.dot(g.double(), v.double()) / N
    assert rel_err(Fv, ref.cpu().numpy()) < 2e-4
    halves = [C.EFLinearOperator(model, lf, params, [(X[i:i + 4], y[i:i + 4])], num_data=N, check_deterministic=False)
              for i in (0, 4)]
    assert rel_err(halves[0] @ v + halves[1] @ v, Fv.double().cpu().numpy()) < 2e-4
    # Hutch++ with 96 products (3 x K = 32 packed probes): against the exact trace sum_n ||g_n||^2 / N
    tr_exact = 0.0
    for n in range(N):
        out = model(X[n:n + 1])
        g = torch.autograd.grad(nn.functional.cross_entropy(out, y[n:n + 1]), plist)
        tr_exact += sum(float(t.double().square().sum()) for t in g) / N
    est = float(C.hutchpp_trace(EF, 96))
    assert abs(est - tr_exact) <= 0.05 * tr_exact  # rank(F) <= 8 << 32: the range part is (almost) exact


@pytest.mark.gpu
@pytest.mark.parametrize("fisher", ["type-2", "mc"])
def test_lenet_c3_full_batch(dev, fisher):
    """BASELINE C3 at its full size: LeNet-5, B = 1024, type-2 (V = 10) and mc: factors vs float64 torch
    on the same draws (type-2 is deterministic; mc compared in expectation through the type-2 factors)."""
    torch.manual_seed(0)
    model = lenet5().to(dev).eval()
    params = dict(model.named_parameters())
    B = 1024
    X, y = torch.rand(B, 1, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
    lf = nn.CrossEntropyLoss()
    kw = dict(separate_weight_and_bias=False, check_deterministic=False)
    K = C.KFACLinearOperator(model, lf, params, [(X, y)], fisher_type=fisher, **kw)
    model64 = lenet5().to(dev).double().eval()
    model64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    K64 = C.KFACLinearOperator(model64, lf, dict(model64.named_parameters()), [(X.double(), y)], fisher_type="type-2", **kw)
    for blk, blk64 in zip(K[1], K64[1]):
        for f, (S, S64) in enumerate(zip(blk, blk64)):
            assert torch.equal(S, S.T)
            if f == 1:  # input covariances: our SYRK on the layer inputs, independent of the Fisher type
                assert rel_err(S, S64.cpu().numpy()) < 1e-4
            elif fisher == "type-2":  # gradient covariances: fp32 autograd (host framework) vs float64
                assert rel_err(S, S64.cpu().numpy()) < 1e-3
            else:  # one MC sample per datum, 1024 data: the gradient covariance in expectation
                assert rel_err(S, S64.cpu().numpy()) < 0.35
    v = torch.rand(K.shape[1], device=dev)
    for kwargs in (dict(), dict(use_heuristic_damping=True), dict(use_exact_damping=True)):
        Kinv = K.inverse(damping=1e-3, **kwargs)
        w = Kinv @ v
        assert torch.isfinite(w).all()
    if fisher == "type-2":
        ref = (K64.inverse(damping=1e-3) @ v.double()).cpu().numpy()
        assert rel_err(K.inverse(damping=1e-3) @ v, ref) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["resnet_toy", "lenet", "resnet18"])
def test_fused_patch_factors_equal_materialised(dev, net, monkeypatch):
    """Input covariances of Conv2d layers with the patches generated inside the SYRK's tile loader
    (`clo_im2col_syrk_accum_f32`, forced for every layer) == the materialised-patch path, <= 2e-5,
    bitwise symmetric; the default policy (fused where it wins) gives the same factors."""
    from curvlinops_amd import computers

    torch.manual_seed(0)
    if net == "resnet_toy":
        model, X, y = ResNetToy().to(dev).eval(), torch.rand(6, 3, 8, 8, device=dev), torch.randint(0, 5, (6,), device=dev)
    elif net == "lenet":
        model, X, y = lenet5().to(dev).eval(), torch.rand(64, 1, 32, 32, device=dev), torch.randint(0, 10, (64,), device=dev)
    else:
        model, X, y = ResNet18().to(dev).eval(), torch.rand(64, 3, 32, 32, device=dev), torch.randint(0, 10, (64,), device=dev)
    params = kfac_params(model)
    kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
    facs = {}
    monkeypatch.setattr(computers, "_CAPTURE", False)
    for mode in ("0", "all", "1", "pixel"):
        monkeypatch.setattr(computers, "_FUSED_IM2COL", "1" if mode == "pixel" else mode)
        monkeypatch.setattr(computers, "_PIXEL_GRAM", mode == "pixel")   # (pixel Gram + fold on small feature maps)
        K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y), (X.flip(0), y.flip(0))], **kw)
        facs[mode] = [blk[1] for blk in K[1] if len(blk) == 2]
    for S0, S1, S2, S3 in zip(facs["0"], facs["all"], facs["1"], facs["pixel"]):
        assert torch.equal(S1, S1.T) and torch.equal(S3, S3.T)
        assert rel_err(S1, S0.double().cpu().numpy()) < 2e-5
        assert rel_err(S2, S0.double().cpu().numpy()) < 2e-5
        assert rel_err(S3, S0.double().cpu().numpy()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["resnet_toy", "lenet"])
@pytest.mark.parametrize("fisher", ["mc", "empirical", "forward-only", "type-2"])
def test_captured_factor_build_equals_eager(dev, net, fisher, monkeypatch):
    """The hipGraph-captured factor build (`computers._CapturedBatch`: the analogue of the reference's traced backend,
    computers/kfac_make_fx.py:26-111) replays the SAME computation: factors equal the eager build's (same MC draws:
    the graph registers the generator), for other data of the captured shape, after `.data` updates of the parameters
    (addresses are baked, values are live), with a ragged last mini-batch (eager) accumulated behind captured ones."""
    from curvlinops_amd import computers

    torch.manual_seed(0)
    if net == "resnet_toy":
        model, mk = ResNetToy().to(dev).eval(), lambda n: (torch.rand(n, 3, 8, 8, device=dev), torch.randint(0, 5, (n,), device=dev))  # noqa: E731
    else:
        model, mk = lenet5().to(dev).eval(), lambda n: (torch.rand(n, 1, 32, 32, device=dev), torch.randint(0, 10, (n,), device=dev))  # noqa: E731
    params = kfac_params(model)
    kw = dict(fisher_type=fisher, separate_weight_and_bias=False, check_deterministic=False)
    computers.reset_captured_builds()
    monkeypatch.setattr(computers, "_CAPTURE_MANUAL", True)

    def factors(data, capture):
        monkeypatch.setattr(computers, "_CAPTURE", capture)
        K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, data, **kw)
        return [f for blk in K[1] for f in blk]

    def same(a, b, tol=2e-5):
        for x, y_ in zip(a, b):
            assert rel_err(x, y_.double().cpu().numpy()) < tol

    d1 = [mk(12), mk(12), mk(5)]
    eager = factors(d1, False)
    first = factors(d1, True)      # first 12-row batch eager (library warm-up), the second one captured + replayed
    second = factors(d1, True)     # the 5-row batch comes back: captured too
    assert sum(isinstance(v, computers._CapturedBatch) for v in computers._CAPTURED.values()) == 2
    third = factors(d1, True)      # pure replay
    same(first, eager), same(second, eager), same(third, eager)
    d2 = [mk(12), mk(12), mk(5)]
    same(factors(d2, True), factors(d2, False))
    for p in params.values():
        p.data.mul_(1.0 + 0.1 * torch.rand_like(p))
    moved, replay = factors(d2, False), factors(d2, True)
    same(replay, moved)
    assert max(rel_err(a, b.double().cpu().numpy()) for a, b in zip(third, replay)) > 1e-3
    computers.reset_captured_builds()


@pytest.mark.gpu
def test_resnet18_fp32_native_against_fp64_torch_path_gpu():
    """BASELINE C4 at full size (128 rows): the float32 native path (HIP factors, Cholesky / eigendecomposition
    post-processing, Kronecker matvecs) against this package's float64 torch path on the same device -- products
    1e-4, damped inverses 1e-3.  Includes the layer4 factor on which the vendor eigensolver loses
    orthogonality (dead ReLU features)."""
    import copy

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m32 = ResNet18().to(dev).eval()
    m64 = copy.deepcopy(m32).double()
    B = 128
    X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
    kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
    lf = nn.CrossEntropyLoss()
    for cls in (C.KFACLinearOperator, C.EKFACLinearOperator):
        K32 = cls(m32, lf, kfac_params(m32), [(X, y)], **kw)
        K64 = cls(m64, lf, kfac_params(m64), [(X.double(), y)], **kw)
        v = torch.rand(K32.shape[1], 2, device=dev) - 0.5
        assert rel_err(K32 @ v, (K64 @ v.double()).cpu().numpy()) < 1e-4
        modes = ({}, {"use_heuristic_damping": True}, {"use_exact_damping": True}) if cls is C.KFACLinearOperator else ({},)
        for mode in modes:
            got = K32.inverse(damping=1e-2, **mode) @ v
            ref = K64.inverse(damping=1e-2, **mode) @ v.double()
            assert rel_err(got, ref.cpu().numpy()) < 1e-3, (cls.__name__, mode)


@pytest.mark.gpu
def test_encoder_fp32_native_against_fp64_torch_path_gpu():
    """BASELINE C5's model at full size (12 layers, d = 768, 85 M parameters; 4 sequences of 32 tokens, weight
    sharing over the sequence): KFAC / EKFAC of the Linear layers, float32 native path against the float64 torch
    path on the same device.  Products 1e-4; the damped inverses see factors of rank 128 in 769 / 3073 dimensions
    (condition number ~1e4 after damping): 1e-3 for the EKFAC inverse (eigenbasis route, measured 3.6e-4), 5e-3 for the
    KFAC inverse (fp32 Cholesky of the damped factors, measured 1.9e-3 -- kappa x eps of the factorisation itself)."""
    import copy

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m32 = Encoder().to(dev).eval()
    m64 = copy.deepcopy(m32).double()
    X, y = torch.rand(4, 32, 768, device=dev), torch.randint(0, 10, (4,), device=dev)
    kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
    lf = nn.CrossEntropyLoss()
    for cls in (C.KFACLinearOperator, C.EKFACLinearOperator):
        K32 = cls(m32, lf, kfac_params(m32), [(X, y)], **kw)
        K64 = cls(m64, lf, kfac_params(m64), [(X.double(), y)], **kw)
        v = torch.rand(K32.shape[1], 2, device=dev) - 0.5
        assert rel_err(K32 @ v, (K64 @ v.double()).cpu().numpy()) < 1e-4
        got, ref = K32.inverse(damping=1e-2) @ v, K64.inverse(damping=1e-2) @ v.double()
        assert rel_err(got, ref.cpu().numpy()) < (1e-3 if cls is C.EKFACLinearOperator else 5e-3), cls.__name__
