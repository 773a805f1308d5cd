"""GPU parity tests through the public operator API (fp32 on the MI355X) against (a) golden
vectors produced by the reference in float64 and (b) size-independent properties at the
BASELINE.json sizes.  Tolerances: products and factors
max|y - y_ref| / max|y_ref| <= 1e-4, damped inverses <= 1e-3."""

import numpy as np
import copy

import pytest
import torch
from torch import nn

import curvlinops_amd as C
from conftest import load_golden, mlp_case_tensors
from helpers import KFAC_MODELS, LOSS, build_mlp, golden_data, load_into, rel_err

pytestmark = pytest.mark.gpu

F32 = torch.float32
TOL, TOL_INV = 1e-4, 1e-3


@pytest.fixture(scope="module")
def dev():
    from curvlinops_amd import _hip

    _hip.load()
    return torch.device("cuda:0")


def g32(x, dev):
    return torch.as_tensor(np.asarray(x), dtype=F32).to(dev)


# ----------------------------------------------------------------------------- curvature ops
@pytest.mark.parametrize("case", sorted(load_golden("mlp_curvature")))
@pytest.mark.parametrize("name,cls,native", [("ggn", C.GGNLinearOperator, True), ("ef", C.EFLinearOperator, True),
                                             ("hessian", C.HessianLinearOperator, True)])
def test_curvature_operators_gpu(dev, golden_mlp, case, name, cls, native):
    rec = golden_mlp[case]
    dims, acts, bias, loss, red, *_ = mlp_case_tensors(rec)
    model = build_mlp(dims, acts, bias)
    params = load_into(model, rec, F32, dev)
    data = golden_data(rec, F32, dev, loss)
    op = cls(model, LOSS[loss](reduction=red), params, data)
    assert op.uses_native_kernels == native   # odd layer widths included (scalar-load kernel variants)
    v, V = g32(rec["v"], dev), g32(rec["V"], dev)
    assert rel_err(op @ v, rec[f"{name}_v"]) < TOL
    assert rel_err(op @ V, rec[f"{name}_V"]) < TOL
    assert rel_err(V.T.contiguous() @ op, rec[f"{name}_V"].T) < TOL
    shapes = [p.shape for p in params.values()]
    vl = [c.reshape(s) for c, s in zip(v.split([s.numel() for s in shapes]), shapes)]
    out = op @ vl
    assert rel_err(torch.cat([o.flatten() for o in out]), rec[f"{name}_v"]) < TOL
    assert rel_err(op.to_scipy() @ rec["v"], rec[f"{name}_v"]) < TOL


@pytest.mark.parametrize("case", sorted(load_golden("mlp_columns")))
@pytest.mark.parametrize("name,cls", [("ggn", C.GGNLinearOperator), ("ef", C.EFLinearOperator),
                                      ("hessian", C.HessianLinearOperator)])
def test_native_column_kernels_vs_reference_golden(dev, case, name, cls):
    """Round 4: K = 8 GGN / EF / exact-Hessian columns through ``clo_mlp_ggn_matmat`` / ``clo_mlp_hessian_matmat``
    (asserted: the column kernels ran) against goldens generated from the REFERENCE's vmap
    (_torch_base.py:946-989 over ggn.py:41-72, gradient_moments.py:48-87, hessian.py:66); fp32 vs float64, 1e-4."""
    rec = load_golden("mlp_columns")[case]
    dims, acts, bias, loss, red, *_ = mlp_case_tensors(rec)
    model = build_mlp(dims, acts, bias)
    params = load_into(model, rec, F32, dev)
    data = golden_data(rec, F32, dev, loss)
    op = cls(model, LOSS[loss](reduction=red), params, data)
    assert op.uses_native_kernels
    V = g32(rec["V"], dev)
    got = op @ V
    assert op.native_column_products >= 1, "the K-column kernels did not run"
    assert rel_err(got, rec[f"{name}_V"]) < TOL
    for k in (0, 7):   # and column by column through the single-vector kernels
        assert rel_err(op @ V[:, k].contiguous(), rec[f"{name}_V"][:, k]) < TOL


def test_native_matches_autograd_path_on_gpu(dev):
    """Same operator, both execution paths, same device and dtype."""
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(300, 500), nn.ReLU(), nn.Linear(500, 260), nn.Tanh(), nn.Linear(260, 12)).to(dev)
    params = dict(model.named_parameters())
    data = [(torch.rand(8, 300, device=dev), torch.randint(0, 12, (8,), device=dev)),
            (torch.rand(5, 300, device=dev), torch.randint(0, 12, (5,), device=dev)),
            (torch.rand(40, 300, device=dev), torch.randint(0, 12, (40,), device=dev))]
    for cls in (C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator):
        nat = cls(model, nn.CrossEntropyLoss(), params, data, check_deterministic=False)
        assert nat.uses_native_kernels
        ref = cls(model, nn.CrossEntropyLoss(), params, data, check_deterministic=False)
        ref._native = None
        V = torch.rand(nat.shape[1], 3, device=dev)
        a, b = nat @ V, ref @ V
        assert rel_err(a, b.double().cpu().numpy()) < 2e-4


@pytest.mark.parametrize("loss_name", ["ce", "mse", "bce"])
def test_native_hessian_column_kernels_gpu(dev, loss_name):
    """`H @ M` with K % 4 == 0 columns runs clo_mlp_hessian_matmat (the R-operator on the K-column pipeline; reference
    `hessian.py:66` under the vmap of `_torch_base.py:946-989`): same result as the torch.func path in float64 and as
    single-vector products, several mini-batches (one of them two 8-row passes), tanh / sigmoid / ReLU layers, K > 64."""
    torch.manual_seed(2)
    model = nn.Sequential(nn.Linear(300, 500), nn.Tanh(), nn.Linear(500, 260), nn.ReLU(), nn.Linear(260, 64),
                          nn.Sigmoid(), nn.Linear(64, 12)).to(dev)
    params = dict(model.named_parameters())
    if loss_name == "ce":
        loss, tgt = nn.CrossEntropyLoss(), lambda n: torch.randint(0, 12, (n,), device=dev)
    elif loss_name == "mse":
        loss, tgt = nn.MSELoss(reduction="sum"), lambda n: torch.rand(n, 12, device=dev)
    else:
        loss, tgt = nn.BCEWithLogitsLoss(), lambda n: torch.randint(0, 2, (n, 12), device=dev).float()
    data = [(torch.rand(8, 300, device=dev), tgt(8)), (torch.rand(5, 300, device=dev), tgt(5)),
            (torch.rand(13, 300, device=dev), tgt(13))]
    nat = C.HessianLinearOperator(model, loss, params, data, check_deterministic=False)
    assert nat.uses_native_kernels
    model64 = copy.deepcopy(model).double()
    data64 = [(X.double(), y.double() if y.is_floating_point() else y) for X, y in data]
    ref = C.HessianLinearOperator(model64, loss, dict(model64.named_parameters()), data64, check_deterministic=False)
    ref._native = None
    for K in (8, 72):
        V = torch.rand(nat.shape[1], K, device=dev) - 0.5
        calls = []
        orig = nat._native.plan.hessian_matmat_ptrs
        nat._native.plan.hessian_matmat_ptrs = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        a = nat @ V
        nat._native.plan.hessian_matmat_ptrs = orig
        assert calls, "the Hessian column kernels must have run"
        Kc = min(K, 8)
        b = ref @ V[:, :Kc].double()
        assert rel_err(a[:, :Kc], b.cpu().numpy()) < 2e-4
        for k in (0, K - 1):   # columns are independent: equal to plain matvecs (the single-vector kernels)
            assert rel_err(a[:, k], (nat @ V[:, k].contiguous()).double().cpu().numpy()) < 2e-4


def test_native_column_kernels_match_autograd_path_on_gpu(dev):
    """`A @ M` with K % 4 == 0 columns runs clo_mlp_ggn_matmat (K-trailing, no transposes); same result
    as the torch.func path, for GGN and EF, several mini-batches (incl. a 2-pass one), K > 64 chunked."""
    torch.manual_seed(1)
    model = nn.Sequential(nn.Linear(300, 500), nn.ReLU(), nn.Linear(500, 260), nn.Tanh(), nn.Linear(260, 12)).to(dev)
    params = dict(model.named_parameters())
    data = [(torch.rand(8, 300, device=dev), torch.randint(0, 12, (8,), device=dev)),
            (torch.rand(5, 300, device=dev), torch.randint(0, 12, (5,), device=dev)),
            (torch.rand(13, 300, device=dev), torch.randint(0, 12, (13,), device=dev))]
    for cls in (C.GGNLinearOperator, C.EFLinearOperator):
        nat = cls(model, nn.CrossEntropyLoss(), params, data, check_deterministic=False)
        ref = cls(model, nn.CrossEntropyLoss(), params, data, check_deterministic=False)
        ref._native = None
        for K in (8, 68):
            V = torch.rand(nat.shape[1], K, device=dev) - 0.5
            calls = []
            orig = nat._native.plan.ggn_matmat_ptrs
            nat._native.plan.ggn_matmat_ptrs = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            a = nat @ V
            nat._native.plan.ggn_matmat_ptrs = orig
            assert calls, "the column kernels must have run"
            Kc = min(K, 8)
            b = ref @ V[:, :Kc].contiguous()
            assert rel_err(a[:, :Kc], b.double().cpu().numpy()) < 2e-4
            # columns are independent: the last one equals a plain matvec
            assert rel_err(a[:, K - 1], (nat @ V[:, K - 1].contiguous()).double().cpu().numpy()) < 2e-4
        # tensor-list format in, tensor-list format out
        V = torch.rand(nat.shape[1], 8, device=dev)
        shapes = [p.shape for p in params.values()]
        Vl = [c.reshape(*s, 8) for c, s in zip(V.split([s.numel() for s in shapes]), shapes)]
        out = nat @ Vl
        assert rel_err(torch.cat([o.reshape(-1, 8) for o in out]), (nat @ V).double().cpu().numpy()) < 1e-6


@pytest.mark.parametrize("loss", ["ce", "mse", "bce"])
def test_native_mc_ggn_matches_autograd_path_on_gpu(dev, loss):
    """MC-GGN (sampled would-be gradients, ggn.py:100-168) on the native kernels: the samples are
    drawn from the same seeded global RNG in the same order, so both paths agree to fp32 accuracy --
    vectors, a K-column block (column kernels) and a batch beyond 8 rows (GEMM engine)."""
    torch.manual_seed(2)
    model = nn.Sequential(nn.Linear(40, 64), nn.Tanh(), nn.Linear(64, 12)).to(dev)
    params = dict(model.named_parameters())
    lossf = {"ce": nn.CrossEntropyLoss(), "mse": nn.MSELoss(), "bce": nn.BCEWithLogitsLoss(reduction="sum")}[loss]
    def target(n):
        return torch.randint(0, 12, (n,), device=dev) if loss == "ce" else torch.rand(n, 12, device=dev)
    data = [(torch.rand(n, 40, device=dev), target(n)) for n in (8, 3, 20)]
    nat = C.GGNLinearOperator(model, lossf, params, data, check_deterministic=False, mc_samples=3, seed=123)
    ref = C.GGNLinearOperator(model, lossf, params, data, check_deterministic=False, mc_samples=3, seed=123)
    assert nat.uses_native_kernels
    ref._native = None
    v = torch.rand(nat.shape[1], device=dev) - 0.5
    assert rel_err(nat @ v, (ref @ v).double().cpu().numpy()) < 2e-4
    V = torch.rand(nat.shape[1], 8, device=dev) - 0.5
    assert rel_err(nat @ V, (ref @ V).double().cpu().numpy()) < 2e-4
    # deterministic for a fixed seed, different for another
    assert torch.equal(nat @ v, nat @ v)
    other = C.GGNLinearOperator(model, lossf, params, data, check_deterministic=False, mc_samples=3, seed=7)
    assert not torch.allclose(other @ v, nat @ v)


class TestC2FullSize:
    """BASELINE.json configs[1]: MLP 1024-2688-2688-10 (D = 10 010 122), MSE, GGN."""

    @pytest.fixture(scope="class")
    def setup(self):
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
        params = dict(model.named_parameters())
        X, y = torch.rand(24, 1024, device=dev), torch.rand(24, 10, device=dev)
        return dev, model, params, X, y

    def test_shape_and_native(self, setup):
        dev, model, params, X, y = setup
        G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X[:8], y[:8])], check_deterministic=False)
        assert G.shape == (10_010_122, 10_010_122) and G.uses_native_kernels

    def test_linearity_symmetry_psd(self, setup):
        dev, model, params, X, y = setup
        G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X[:8], y[:8])], check_deterministic=False)
        D = G.shape[1]
        v, w = torch.rand(D, device=dev) - 0.5, torch.rand(D, device=dev) - 0.5
        Gv, Gw = G @ v, G @ w
        lin = G @ (2.0 * v - 3.0 * w)
        assert rel_err(lin, (2.0 * Gv - 3.0 * Gw).double().cpu().numpy()) < 1e-4
        vGw, wGv = torch.dot(v.double(), Gw.double()), torch.dot(w.double(), Gv.double())
        assert abs(vGw - wGv) / max(abs(vGw), 1e-30) < 1e-3
        assert torch.dot(v.double(), Gv.double()) >= 0

    def test_batch_split_invariance(self, setup):
        """sum over mini-batches with B_b / N_data weights == one big batch (mean reduction);
        also exercises the 8-row, 16-row and MFMA (N > 16) kernel paths."""
        dev, model, params, X, y = setup
        one = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
        split = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X[:8], y[:8]), (X[8:11], y[8:11]), (X[11:], y[11:])],
                                    check_deterministic=False)
        v = torch.rand(one.shape[1], device=dev)
        assert rel_err(split @ v, (one @ v).double().cpu().numpy()) < 2e-4

    def test_columns_vs_matvecs(self, setup):
        """K = 8 columns at full size through the K-trailing kernels == 8 matvecs."""
        dev, model, params, X, y = setup
        G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X[:8], y[:8]), (X[8:11], y[8:11])],
                                check_deterministic=False)
        V = torch.rand(G.shape[1], 8, device=dev) - 0.5
        GV = G @ V
        for k in (0, 5, 7):
            assert rel_err(GV[:, k], (G @ V[:, k].contiguous()).double().cpu().numpy()) < 2e-4

    def test_against_autograd_path(self, setup):
        dev, model, params, X, y = setup
        data = [(X[:8], y[:8])]
        nat = C.GGNLinearOperator(model, nn.MSELoss(), params, data, check_deterministic=False)
        ref = C.GGNLinearOperator(model, nn.MSELoss(), params, data, check_deterministic=False)
        ref._native = None
        v = torch.rand(nat.shape[1], device=dev)
        assert rel_err(nat @ v, (ref @ v).double().cpu().numpy()) < 2e-4


    @pytest.mark.parametrize("N,cls,loss", [(13, "ggn", "mse"), (40, "ggn", "ce"), (64, "ef", "mse"), (48, "ggn", "mse"),
                                            (100, "ggn", "ce")])
    def test_row_regimes_against_autograd(self, setup, N, cls, loss):
        """Full-size net at the row counts of every kernel regime (9-16, 33-48, 49-64 rows on the MFMA streaming
        chain incl. its slab-free first-layer kernel, > 64 on the GEMM engine) against the torch.func path."""
        dev, model, params, _, _ = setup
        torch.manual_seed(N)
        X = torch.rand(N, 1024, device=dev)
        if loss == "ce":
            y, lf = torch.randint(0, 10, (N,), device=dev), nn.CrossEntropyLoss()
        else:
            y, lf = torch.rand(N, 10, device=dev), nn.MSELoss()
        Op = C.GGNLinearOperator if cls == "ggn" else C.EFLinearOperator
        nat = Op(model, lf, params, [(X, y)], check_deterministic=False)
        ref = Op(model, lf, params, [(X, y)], check_deterministic=False)
        assert nat.uses_native_kernels
        ref._native = None
        v = torch.rand(nat.shape[1], device=dev) - 0.5
        assert rel_err(nat @ v, (ref @ v).double().cpu().numpy()) < 2e-4


# ----------------------------------------------------------------------------- structured ops
@pytest.mark.parametrize("name", ["rect", "sq", "one", "three"])
def test_kronecker_gpu(dev, golden_linops, name):
    rec = golden_linops[f"kron_{name}"]
    fs = [g32(rec[f"factor{i}"], dev) for i in range(sum(k.startswith("factor") for k in rec))]
    K = C.KroneckerProductLinearOperator(*fs)
    X, Y = g32(rec["X"], dev), g32(rec["Y"], dev)
    assert rel_err(K @ X, rec["KX"]) < TOL
    assert rel_err(K @ X[:, 0].contiguous(), rec["KX"][:, 0]) < TOL
    assert rel_err(K.adjoint() @ Y, rec["KTY"]) < TOL
    if name in ("sq", "one"):
        assert rel_err(K.inverse(damping=1e-2) @ X, rec["inv_plain_X"]) < TOL_INV
        assert rel_err(K.inverse(damping=1e-2, use_exact_damping=True) @ X, rec["inv_exact_X"]) < TOL_INV
        if "inv_heur_X" in rec:
            got = K.inverse(damping=1e-2, use_heuristic_damping=True, min_damping=1e-3) @ X
            assert rel_err(got, rec["inv_heur_X"]) < TOL_INV


def test_kronecker_large_gpu(dev):
    """C2-sized joint block: G 2688^2 (x) A 2689^2 against torch on the same device (fp64)."""
    torch.manual_seed(1)
    G = torch.rand(2688, 2688, device=dev) / 2688
    A = torch.rand(2689, 2689, device=dev) / 2689
    K = C.KroneckerProductLinearOperator(G, A)
    for cols in (1, 3):
        X = torch.rand(2688 * 2689, cols, device=dev)
        ref = torch.einsum("Aa,abz,Bb->ABz", G.double(), X.double().view(2688, 2689, cols), A.double()).reshape(-1, cols)
        assert rel_err(K @ X, ref.cpu().numpy()) < TOL


def test_eigh_operator_gpu(dev, golden_linops):
    rec = golden_linops["eigh"]
    E = C.EighDecomposedLinearOperator(g32(rec["lam"], dev),
                                       C.KroneckerProductLinearOperator(g32(rec["Q1"], dev), g32(rec["Q2"], dev)))
    X = g32(rec["X"], dev)
    assert rel_err(E @ X, rec["EX"]) < TOL
    assert rel_err(E.inverse(damping=0.05) @ X, rec["invEX"]) < TOL
    E2 = C.EighDecomposedLinearOperator(g32(rec["lam"], dev), torch.kron(g32(rec["Q1"], dev), g32(rec["Q2"], dev)))
    assert rel_err(E2 @ X, rec["EX"]) < TOL


def test_gemm_sqsum_kernel(dev):
    from curvlinops_amd import _hip

    g = torch.Generator().manual_seed(3)
    for nb, M, K, N in ((1, 5, 3, 4), (7, 64, 16, 130), (33, 130, 1, 129), (200, 20, 49, 17)):
        A = torch.rand(nb, M, K, generator=g, dtype=torch.float64) - 0.5
        B = torch.rand(nb, K, N, generator=g, dtype=torch.float64) - 0.5
        C0 = torch.rand(M, N, generator=g, dtype=torch.float64)
        out = C0.float().to(dev)
        _hip.gemm_sqsum(A.float().to(dev), B.float().to(dev), out, alpha=0.5, beta=2.0)
        ref = 0.5 * (A @ B).square().sum(0) + 2.0 * C0
        assert rel_err(out, ref.numpy()) < 2e-5
        # transposed (strided) A operand as used by the EKFAC correction
        At = A.transpose(1, 2).contiguous().float().to(dev)
        out = torch.zeros(M, N, device=dev)
        _hip.gemm_sqsum(At.transpose(1, 2), B.float().to(dev), out)
        assert rel_err(out, (A @ B).square().sum(0).numpy()) < 2e-5


# ----------------------------------------------------------------------------- KFAC / EKFAC
@pytest.mark.parametrize("case", sorted(load_golden("kfac")))
def test_kfac_gpu(dev, case):
    rec = load_golden("kfac")[case]
    loss, red = str(rec["loss"]), str(rec["reduction"])
    model = KFAC_MODELS[case]()
    params = load_into(model, rec, F32, dev)
    data = golden_data(rec, F32, dev, loss)
    V = g32(rec["V"], dev)
    for tag in sorted({k.split("/")[0] for k in rec if "|" in k and not k.startswith("ekfac")}):
        fisher, approx, sep = tag.split("|")
        K = C.KFACLinearOperator(model, LOSS[loss](reduction=red), params, data, fisher_type=fisher,
                                 kfac_approx=approx, separate_weight_and_bias=sep == "sep")
        _, Kc, _ = K
        for b, block in enumerate(Kc):
            for f, fac in enumerate(block):
                assert rel_err(fac, rec[f"{tag}/block{b}_factor{f}"]) < TOL, (case, tag, b, f)
        assert rel_err(K @ V, rec[f"{tag}/KV"]) < TOL
        assert rel_err(K.trace(), rec[f"{tag}/trace"]) < TOL
        assert rel_err(K.inverse(damping=1e-2) @ V, rec[f"{tag}/inv_plain"]) < TOL_INV
        assert rel_err(K.inverse(damping=1e-2, use_exact_damping=True) @ V, rec[f"{tag}/inv_exact"]) < TOL_INV
        if f"{tag}/inv_heur" in rec:
            got = K.inverse(damping=1e-2, use_heuristic_damping=True, min_damping=1e-4) @ V
            assert rel_err(got, rec[f"{tag}/inv_heur"]) < TOL_INV
    assert all(p.grad is None for p in model.parameters())


@pytest.mark.parametrize("case", [c for c in sorted(load_golden("kfac")) if not c.startswith("seq")])
def test_ekfac_gpu(dev, case):
    rec = load_golden("kfac")[case]
    loss, red = str(rec["loss"]), str(rec["reduction"])
    model = KFAC_MODELS[case]()
    params = load_into(model, rec, F32, dev)
    data = golden_data(rec, F32, dev, loss)
    V = g32(rec["V"], dev)
    for tag in sorted({k.split("/")[0] for k in rec if k.startswith("ekfac")}):
        _, fisher, sep = tag.split("|")
        E = C.EKFACLinearOperator(model, LOSS[loss](reduction=red), params, data, fisher_type=fisher,
                                  separate_weight_and_bias=sep == "sep")
        assert rel_err(E.trace(), rec[f"{tag}/trace"]) < TOL_INV
        # the product and the damped inverse product are basis-independent quantities (the eigenbases
        # themselves are not unique): both against the reference's float64 values
        assert rel_err(E @ V, rec[f"{tag}/EV"]) < TOL_INV, (case, tag)
        assert rel_err(E.inverse(damping=1e-2) @ V, rec[f"{tag}/invEV"]) < TOL_INV, (case, tag)


def test_kfac_factor_properties_lenet_size(dev):
    """C3-like: LeNet-5 on 32x32 inputs, B = 256: factors symmetric PSD, equal to the float64
    torch result on the same device for every layer (conv via unfold, joint bias column)."""
    torch.manual_seed(0)
    model = nn.Sequential(
        nn.Conv2d(1, 6, 5), nn.ReLU(), nn.MaxPool2d(2), nn.Conv2d(6, 16, 5), nn.ReLU(), nn.MaxPool2d(2),
        nn.Flatten(), nn.Linear(400, 120), nn.ReLU(), nn.Linear(120, 84), nn.ReLU(), nn.Linear(84, 10),
    ).to(dev)
    params = dict(model.named_parameters())
    X, y = torch.rand(256, 1, 32, 32, device=dev), torch.randint(0, 10, (256,), device=dev)
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="empirical",
                             separate_weight_and_bias=False, check_deterministic=False)
    m64 = nn.Sequential(
        nn.Conv2d(1, 6, 5), nn.ReLU(), nn.MaxPool2d(2), nn.Conv2d(6, 16, 5), nn.ReLU(), nn.MaxPool2d(2),
        nn.Flatten(), nn.Linear(400, 120), nn.ReLU(), nn.Linear(120, 84), nn.ReLU(), nn.Linear(84, 10),
    ).to(dev).double()
    m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    K64 = C.KFACLinearOperator(m64, nn.CrossEntropyLoss(), dict(m64.named_parameters()), [(X.double(), y)],
                               fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
    for blk, blk64 in zip(K[1], K64[1]):
        for f, f64 in zip(blk, blk64):
            assert torch.equal(f, f.T)
            assert rel_err(f, f64.cpu().numpy()) < TOL
    v = torch.rand(K.shape[1], device=dev)
    assert rel_err(K @ v, (K64 @ v.double()).cpu().numpy()) < TOL
    assert rel_err(K.inverse(damping=1e-3) @ v, (K64.inverse(damping=1e-3) @ v.double()).cpu().numpy()) < TOL_INV


# ----------------------------------------------------------------------------- trace
def test_trace_estimators_gpu(dev):
    rec = load_golden("trace")["t"]
    A = g32(rec["A"], dev)
    op = C.KroneckerProductLinearOperator(A)  # a dense symmetric operator
    for dist in ("rademacher", "normal"):
        pool = g32(rec[f"{dist}/pool"], dev)
        assert rel_err(C.hutchinson_trace(op, 12, dist, probes=pool[:, :12].contiguous()), rec[f"{dist}/hutch"]) < TOL
        got = C.hutchpp_trace(op, 24, dist, probes=(pool[:, :8].contiguous(), pool[:, 8:16].contiguous()))
        assert rel_err(got, rec[f"{dist}/hutchpp"]) < 1e-3
        p12 = pool[:, :12].contiguous()
        assert rel_err(C.hutchinson_diag(op, 12, dist, probes=p12), rec[f"{dist}/hutch_diag"]) < TOL
        assert rel_err(C.hutchinson_squared_fro(op, 12, dist, probes=p12), rec[f"{dist}/hutch_fro2"]) < TOL
        # XTrace / XDiag (reference `trace/epperly2024xtrace.py:15-101`, `diagonal/epperly2024xtrace.py`) with the
        # replayed probes: fp32 on the device against the reference's float64 values
        p8 = pool[:, :8].contiguous()
        assert rel_err(C.xtrace(op, 16, dist, probes=p8), rec[f"{dist}/xtrace"]) < 1e-3
        if dist == "rademacher":
            assert rel_err(C.xdiag(op, 16, probes=p8), rec[f"{dist}/xdiag"]) < 1e-3
    torch.manual_seed(0)
    ests = torch.stack([C.hutchinson_trace(op, 29) for _ in range(200)])
    assert abs(ests.mean() - A.trace()) / A.trace() < 0.05


def test_trace_estimators_decaying_spectrum_gpu(dev):
    """Hutch++ / XTrace on an operator whose spectrum decays over six decades INSIDE the sketch (golden from the reference,
    `oracle/make_golden_kfac.py::gen_trace_decay`, injected probes): the fp32 device path keeps all n columns of the range
    basis (`trace._gram_orthonormal_basis`: float64-accumulated Gram passes) like the reference's Householder Q
    (`meyer2020hutch.py:89-93`), so the estimates agree to 1e-4 -- the round-5 basis, which dropped directions below 3e-3
    of the largest, treated ~1e-2 of this trace stochastically.  Also the pieces: an orthonormal n-column basis for a
    rank-deficient block, and `project_out` against float64."""
    from curvlinops_amd import trace as T

    rec = load_golden("trace_decay")["t"]
    U, lam = torch.from_numpy(rec["U"]), torch.from_numpy(rec["lam"])
    A = ((U * lam) @ U.T).float().to(dev)
    op = C.KroneckerProductLinearOperator(A)
    N = 32
    for dist in ("rademacher", "normal"):
        pool = g32(rec[f"{dist}/pool"], dev)
        got = C.hutchpp_trace(op, 3 * N, dist, probes=(pool[:, :N].contiguous(), pool[:, N:2 * N].contiguous()))
        assert rel_err(got, rec[f"{dist}/hutchpp"]) < 1e-4
        assert rel_err(C.xtrace(op, 2 * N, dist, probes=pool[:, :N].contiguous()), rec[f"{dist}/xtrace"]) < 1e-4
    # the basis itself: n orthonormal columns spanning the block, also when the block is rank-deficient
    g = torch.Generator().manual_seed(3)
    Y = torch.randn(70000, 24, generator=g).to(dev) * torch.logspace(0, -6, 24, device=dev)
    Y[:, 5] = 0.0                       # an exactly dependent column
    Y[:, 9] = Y[:, 2]
    Q = T.orthonormal_basis(Y)
    assert Q.shape == Y.shape
    assert float((Q.double().T @ Q.double() - torch.eye(24, device=dev, dtype=torch.float64)).abs().max()) < 1e-5
    resid = Y.double() - Q.double() @ (Q.double().T @ Y.double())
    assert float(resid.abs().max() / Y.abs().max()) < 1e-5      # range(Y) is inside range(Q)
    G = torch.randn(70000, 12, generator=g).to(dev)
    want = G.double() - Q.double() @ (Q.double().T @ G.double())
    assert rel_err(T.project_out(Q, G), want.cpu().numpy()) < 1e-5


def test_tall_gram_and_apply_kernels(dev):
    """`clo_tall_gram_f64` (exact products, float64 accumulation) and `clo_tall_apply_f32` against float64, ragged row and
    column counts, strided views; the Gram is accurate far below float32 resolution."""
    from curvlinops_amd import _hip

    g = torch.Generator().manual_seed(8)
    for m, n1, n2 in ((1, 4, 4), (37, 5, 3), (1000, 16, 16), (4099, 32, 20), (250001, 32, 32), (33333, 64, 48), (5000, 17, 64)):
        X = torch.randn(m, n1 + 3, generator=g).to(dev)[:, :n1]
        Y = torch.randn(m, n2, generator=g).to(dev)
        ref = (X.double().T @ Y.double()).cpu()
        got = _hip.tall_gram(X, Y).cpu()
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-12, (m, n1, n2)
        sym = _hip.tall_gram(X).cpu()
        refs = (X.double().T @ X.double()).cpu()
        assert float((sym - refs).abs().max() / refs.abs().max()) < 1e-12 and torch.equal(sym, sym.T)
    for m, k, n in ((1, 4, 4), (37, 8, 3), (1000, 16, 16), (4099, 32, 20), (250001, 32, 32), (33333, 64, 48), (777, 12, 64)):
        Qm = torch.randn(m, k + 4, generator=g).to(dev)[:, :k]
        Cm = torch.randn(k, n, generator=g).to(dev)
        Gm = torch.randn(m, n, generator=g).to(dev)
        assert _hip.tall_apply_supported(Qm, Cm, Gm)
        ref = (0.5 * Gm.double() + Qm.double() @ Cm.double()).cpu().numpy()
        assert rel_err(_hip.tall_apply(Qm, Cm, Gm, beta=0.5), ref) < 2e-6
        assert rel_err(_hip.tall_apply(Qm, Cm), (Qm.double() @ Cm.double()).cpu().numpy()) < 2e-6


def test_repeated_products_stress(dev):
    """The split-K / row-range slabs and the activation workspace are reused by every product:
    300 back-to-back products with changing vectors, each checked through linearity against
    three reference products (catches stale-buffer / ordering bugs between launches)."""
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
    params = dict(model.named_parameters())
    X, y = torch.rand(8, 1024, device=dev), torch.rand(8, 10, device=dev)
    G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
    D = G.shape[1]
    basis = [torch.rand(D, device=dev) - 0.5 for _ in range(3)]
    Gb = [G @ b for b in basis]
    torch.cuda.synchronize()
    worst = 0.0
    for it in range(300):
        c = torch.rand(3, device=dev) - 0.5
        v = c[0] * basis[0] + c[1] * basis[1] + c[2] * basis[2]
        ref = c[0] * Gb[0] + c[1] * Gb[1] + c[2] * Gb[2]
        got = G @ v
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, err)
    assert worst < 5e-4, worst


def test_gram_orthonormal_basis_gpu(dev):
    """The Gram route used for very tall fp32 blocks: orthonormal to fp32 accuracy, spans range(X)
    (full-rank and rank-deficient), and Hutch++ built on it agrees with the float64 reference."""
    import curvlinops_amd.trace as T

    g = torch.Generator().manual_seed(0)
    X = (torch.rand(300_000, 16, generator=g, dtype=torch.float64) - 0.5) * torch.logspace(0, -2, 16, dtype=torch.float64)
    Xd = X.float().to(dev)
    Q = T.orthonormal_basis(Xd)
    assert Q.shape == (300_000, 16)
    assert (Q.T.double() @ Q.double() - torch.eye(16, device=dev, dtype=torch.float64)).abs().max() < 1e-5
    resid = Xd.double() - Q.double() @ (Q.T.double() @ Xd.double())
    assert resid.abs().max() / Xd.abs().max() < 1e-4
    Xr = (X[:, :3] @ torch.rand(3, 16, generator=g, dtype=torch.float64)).float().to(dev)  # rank 3
    Qr = T.orthonormal_basis(Xr, complete=False)       # rank-revealing form (what Hutch++ uses): the 3 range directions
    assert Qr.shape[1] == 3
    assert (Xr.double() - Qr.double() @ (Qr.T.double() @ Xr.double())).abs().max() / Xr.abs().max() < 1e-4
    Qc = T.orthonormal_basis(Xr)                       # n orthonormal columns, as the reference's Householder Q
    assert Qc.shape[1] == 16
    assert (Qc.T.double() @ Qc.double() - torch.eye(16, device=dev, dtype=torch.float64)).abs().max() < 1e-5
    assert (Xr.double() - Qc.double() @ (Qc.T.double() @ Xr.double())).abs().max() / Xr.abs().max() < 1e-4
    # Hutch++ on a dense PSD matrix: same estimate as float64 torch with the same probes
    n = 300_000 // 64
    B = torch.rand(n, 40, generator=g, dtype=torch.float64)
    A = B @ B.T
    S, G = torch.rand(n, 8, generator=g, dtype=torch.float64) - 0.5, torch.rand(n, 8, generator=g, dtype=torch.float64) - 0.5
    ref = T.hutchpp_trace(A, 24, probes=(S, G))
    got = T.hutchpp_trace(A.float().to(dev), 24, probes=(S.float().to(dev), G.float().to(dev)))
    assert abs(float(got) - float(ref)) / abs(float(ref)) < 1e-3



def test_randomised_native_vs_autograd(dev):
    """40 random MLPs (1-4 layers, aligned and odd widths, all activations / losses / reductions,
    1-3 mini-batches of 1..70 rows, vectors and K-column blocks): native kernels == torch.func path."""
    import os, sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_native

    worst, failures = fuzz_native.run(seed=11, ncase=40)
    assert not failures, failures
    assert worst < 1e-4


@pytest.mark.parametrize("sep", [True, False], ids=["sep", "joint"])
@pytest.mark.parametrize("cls_name", ["KFACLinearOperator", "EKFACLinearOperator"])
def test_equal_shape_blocks_run_batched(dev, cls_name, sep):
    """Repeated layer shapes: the block-diagonal product groups equal-shape Kronecker / eigenbasis
    blocks into batched GEMMs (single vectors); the result must equal the float64 CPU operator, for
    the operator, its damped inverse, and K-column blocks (ungrouped path)."""
    import curvlinops_amd as C
    from curvlinops_amd.kronecker import BlockDiagonalLinearOperator

    torch.manual_seed(0)
    layers = []
    for _ in range(4):
        layers += [nn.Linear(16, 16), nn.Tanh()]
    model64 = nn.Sequential(*layers, nn.Linear(16, 4)).double()
    X64, y64 = torch.rand(12, 16, dtype=torch.float64), torch.randint(0, 4, (12,))
    cls = getattr(C, cls_name)
    kw = dict(fisher_type="type-2", separate_weight_and_bias=sep, check_deterministic=False)
    ref = cls(model64, nn.CrossEntropyLoss(), dict(model64.named_parameters()), [(X64, y64)], **kw)
    model = nn.Sequential(*[type(m)(*((m.in_features, m.out_features) if isinstance(m, nn.Linear) else ())) for m in model64]).to(dev)
    model.load_state_dict({k: v.float() for k, v in model64.state_dict().items()})
    op = cls(model, nn.CrossEntropyLoss(), dict(model.named_parameters()), [(X64.float().to(dev), y64.to(dev))], **kw)
    _, K, _ = op
    assert isinstance(K, BlockDiagonalLinearOperator) and K._kron_groups(), "equal-shape blocks must be grouped"
    v = torch.rand(op.shape[1], dtype=torch.float64)
    V = torch.rand(op.shape[1], 3, dtype=torch.float64)
    assert rel_err((op @ v.float().to(dev)).cpu(), (ref @ v).numpy()) < 1e-4   # batched products on the factors in place
    assert rel_err((op @ V.float().to(dev)).cpu(), (ref @ V).numpy()) < 1e-4
    assert rel_err((op.inverse(damping=1e-1) @ v.float().to(dev)).cpu(), (ref.inverse(damping=1e-1) @ v).numpy()) < 1e-3


@pytest.mark.parametrize("cls_name,loss,red", [
    ("GGNLinearOperator", "mse", "mean"), ("GGNLinearOperator", "ce", "sum"),
    ("EFLinearOperator", "ce", "mean"), ("HessianLinearOperator", "mse", "mean"),
    ("HessianLinearOperator", "ce", "mean"),
])
def test_consecutive_mini_batches_are_merged(dev, cls_name, loss, red):
    """The native path processes consecutive mini-batches as one larger batch where that is cheaper
    (every row carries the same weight scale * B / N_data): same product as batch by batch, and as
    the float64 CPU operator; unequal batch sizes, vectors through `@` (flat path) and matrices."""
    import curvlinops_amd as C

    torch.manual_seed(1)
    model64 = nn.Sequential(nn.Linear(20, 32), nn.Tanh(), nn.Linear(32, 24), nn.ReLU(), nn.Linear(24, 4)).double()
    sizes = (12, 12, 40, 7, 64)
    data64 = []
    for b in sizes:
        X = torch.rand(b, 20, dtype=torch.float64)
        y = torch.randint(0, 4, (b,)) if loss == "ce" else torch.rand(b, 4, dtype=torch.float64)
        data64.append((X, y))
    lf = LOSS[loss](reduction=red)
    cls = getattr(C, cls_name)
    ref = cls(model64, lf, dict(model64.named_parameters()), data64)
    model = nn.Sequential(nn.Linear(20, 32), nn.Tanh(), nn.Linear(32, 24), nn.ReLU(), nn.Linear(24, 4)).to(dev)
    model.load_state_dict({k: v.float() for k, v in model64.state_dict().items()})
    data = [(X.float().to(dev), (y if loss == "ce" else y.float()).to(dev)) for X, y in data64]
    params = dict(model.named_parameters())
    merged = cls(model, lf, params, data, check_deterministic=False)
    single = cls(model, lf, params, data, check_deterministic=False)
    single._MERGE_MAX_ROWS = 0  # batch by batch
    assert merged.uses_native_kernels
    entries = [(X, 0, 1.0, None, 1.0) for X, _ in data]
    assert len(merged._merge_native_batches(entries)) < len(entries)
    assert len(single._merge_native_batches(entries)) == len(entries)
    v64 = torch.rand(ref.shape[1], dtype=torch.float64)
    V64 = torch.rand(ref.shape[1], 3, dtype=torch.float64)
    v, V = v64.float().to(dev), V64.float().to(dev)
    assert rel_err((merged @ v).cpu(), (single @ v).cpu().numpy()) < 1e-5
    assert rel_err((merged @ v).cpu(), (ref @ v64).numpy()) < 1e-4
    assert rel_err((merged @ V).cpu(), (ref @ V64).numpy()) < 1e-4


def test_empty_and_ragged_mini_batches(dev):
    """An empty mini-batch contributes nothing to the sum over data (and must not break the native
    path); ragged batch sizes incl. a single row."""
    import curvlinops_amd as C

    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(8, 12), nn.ReLU(), nn.Linear(12, 4)).to(dev)
    params = dict(model.named_parameters())
    mk = lambda n: (torch.rand(n, 8, device=dev), torch.rand(n, 4, device=dev))  # noqa: E731
    b5, b0, b3, b1 = mk(5), mk(0), mk(3), mk(1)
    for cls in (C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator):
        for red in ("mean", "sum"):
            with_empty = cls(model, nn.MSELoss(reduction=red), params, [b5, b0, b3, b1], check_deterministic=False, num_data=9)
            without = cls(model, nn.MSELoss(reduction=red), params, [b5, b3, b1], check_deterministic=False, num_data=9)
            assert with_empty.uses_native_kernels
            v, V = torch.rand(with_empty.shape[1], device=dev), torch.rand(with_empty.shape[1], 2, device=dev)
            assert rel_err((with_empty @ v).cpu(), (without @ v).cpu().numpy()) < 1e-6
            assert rel_err((with_empty @ V).cpu(), (without @ V).cpu().numpy()) < 1e-6


# ----------------------------------------------------------------------------- round-2 regressions
def test_shared_module_instance_is_not_run_natively(dev):
    """`act = ReLU(); Sequential(Linear, act, Linear, act, Linear)`: `named_children` would drop the
    second activation.  Such a model must take the autograd path and agree with a model that
    uses two activation instances (which runs natively)."""
    torch.manual_seed(0)
    act = nn.ReLU()
    shared = nn.Sequential(nn.Linear(12, 16), act, nn.Linear(16, 8), act, nn.Linear(8, 3)).to(dev)
    twin = nn.Sequential(nn.Linear(12, 16), nn.ReLU(), nn.Linear(16, 8), nn.ReLU(), nn.Linear(8, 3)).to(dev)
    twin.load_state_dict(shared.state_dict())
    X, y = torch.rand(6, 12, device=dev), torch.rand(6, 3, device=dev)
    ops = [C.GGNLinearOperator(m, nn.MSELoss(), dict(m.named_parameters()), [(X, y)]) for m in (shared, twin)]
    assert not ops[0].uses_native_kernels and ops[1].uses_native_kernels
    v = torch.rand(ops[0].shape[1], device=dev)
    assert rel_err(ops[0] @ v, (ops[1] @ v).cpu().numpy()) < TOL


@pytest.mark.parametrize("cls", [C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator])
@pytest.mark.parametrize("loss", ["mse", "ce"])
def test_inplace_updates_of_params_and_data_are_seen(dev, cls, loss):
    """The reference recomputes per product and holds params / data by reference
    (`_torch_base.py:832-905`): after `p.add_()` / `X.mul_()` an existing operator must equal a freshly
    built one (the native path keeps per-batch output gradients, validated by version counters)."""
    torch.manual_seed(1)
    model = build_mlp([16, 24, 12, 4], ["tanh", "relu", "identity"], [True, True, True]).to(dev)
    params = dict(model.named_parameters())
    mk = lambda: (torch.rand(5, 16, device=dev),  # noqa: E731
                  torch.randint(0, 4, (5,), device=dev) if loss == "ce" else torch.rand(5, 4, device=dev))
    data = [mk(), mk()]
    lf = LOSS[loss]()
    op = cls(model, lf, params, data)
    assert op.uses_native_kernels
    v = torch.rand(op.shape[1], device=dev)
    V = torch.rand(op.shape[1], 8, device=dev)
    first = op @ v

    def fresh():
        return cls(model, lf, params, data)

    with torch.no_grad():
        for p in params.values():
            p.add_(0.05 * torch.randn_like(p))
    assert rel_err(first, (fresh() @ v).cpu().numpy()) > 1e-3  # the update matters ...
    assert rel_err(op @ v, (fresh() @ v).cpu().numpy()) < 1e-6  # ... and is seen (flat fast path)
    assert rel_err(op @ V, (fresh() @ V).cpu().numpy()) < 1e-6  # (matrix path)
    with torch.no_grad():
        data[0][0].mul_(1.5)
    assert rel_err(op @ v, (fresh() @ v).cpu().numpy()) < 1e-6
    data[1] = mk()  # a batch replaced by other tensors
    assert rel_err(op @ v, (fresh() @ v).cpu().numpy()) < 1e-6
    assert rel_err(op @ V, (fresh() @ V).cpu().numpy()) < 1e-6
    # a parameter tensor replaced in the dict the operator holds by reference
    name = next(iter(params))
    new = torch.nn.Parameter(params[name].detach() * 0.5)
    setattr(model[0], name.split(".")[1], new)
    params[name] = new
    assert rel_err(op @ v, (fresh() @ v).cpu().numpy()) < 1e-6


@pytest.mark.parametrize("cls", [C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator])
@pytest.mark.parametrize("loss", ["mse", "ce"])
def test_data_style_updates_of_params_and_data_are_seen(dev, cls, loss):
    """Updates that neither replace the ``Parameter`` object nor bump its autograd version -- ``p.data.add_()``,
    ``p.data.copy_()``, ``p.data = t``, ``vector_to_parameters``, ``model.to(dtype).to(float32)``, ``X.data.mul_()`` --
    are what hand-written SGD loops and older optimizers do.  The reference re-reads parameters and data on every
    product (``_torch_base.py:923-944``, ``gradient_moments.py:48-87``, ``hessian.py:66``); an existing operator must
    therefore equal a freshly built one after each of them, in the flat, K-column and list formats."""
    from torch.nn.utils import parameters_to_vector, vector_to_parameters

    torch.manual_seed(2)
    model = build_mlp([16, 24, 12, 4], ["tanh", "relu", "identity"], [True, True, True]).to(dev)
    params = dict(model.named_parameters())
    mk = lambda n: (torch.rand(n, 16, device=dev),  # noqa: E731
                    torch.randint(0, 4, (n,), device=dev) if loss == "ce" else torch.rand(n, 4, device=dev))
    data = [mk(5), mk(5), mk(7)]
    lf = LOSS[loss]()
    op = cls(model, lf, params, data)
    assert op.uses_native_kernels
    v = torch.rand(op.shape[1], device=dev)
    V = torch.rand(op.shape[1], 8, device=dev)
    vl = [torch.rand_like(p) for p in params.values()]

    def check(what):
        new = cls(model, lf, params, data)
        assert op.uses_native_kernels and new.uses_native_kernels, what
        assert rel_err(op @ v, (new @ v).cpu().numpy()) < 1e-6, what          # flat fast path
        assert rel_err(op @ V, (new @ V).cpu().numpy()) < 1e-6, what          # K = 8 columns
        for a, b in zip(op @ vl, new @ vl):                                   # tensor-list format
            assert rel_err(a, b.cpu().numpy()) < 1e-6, what

    first = op @ v
    versions = [p._version for p in params.values()]
    for p in params.values():
        p.data.add_(0.05 * torch.randn_like(p))
    assert [p._version for p in params.values()] == versions          # invisible to the version counters ...
    assert rel_err(first, (cls(model, lf, params, data) @ v).cpu().numpy()) > 1e-3  # ... but it matters
    check("p.data.add_")
    for p in params.values():
        p.data.copy_(p.data * 0.9 + 0.01)
    check("p.data.copy_")
    for p in params.values():
        p.data = (p.data * 1.1).clone()                               # same Parameter object, other storage
    check("p.data = t")
    vector_to_parameters(parameters_to_vector(params.values()) * 0.8, params.values())
    check("vector_to_parameters")
    model.to(torch.float64).to(torch.float32)                        # storage swapped twice, objects kept
    check("model.to(dtype).to(float32)")
    data[0][0].data.mul_(1.3)
    if loss != "ce":
        data[1][1].data.add_(0.2)
    check("X.data.mul_ / y.data.add_")
    data[2][0].data = torch.rand(7, 16, device=dev)
    check("X.data = t")
    # opt-in caching keeps what it derived until refresh()
    op.assume_frozen = True
    kept = op @ v
    for p in params.values():
        p.data.mul_(1.05)
    if cls is C.HessianLinearOperator:   # keeps the per-batch output gradients of the OLD parameters (the EF forms them in-kernel)
        assert rel_err(op @ v, (cls(model, lf, params, data) @ v).cpu().numpy()) > 1e-6
    op.refresh()
    assert rel_err(op @ v, (cls(model, lf, params, data) @ v).cpu().numpy()) < 1e-6
    assert rel_err(kept, (cls(model, lf, params, data) @ v).cpu().numpy()) > 1e-4
    op.assume_frozen = False
    check("after assume_frozen = False")


def test_native_path_is_left_when_params_stop_qualifying(dev):
    """``model.double()`` swaps the parameters' storage for float64 tensors: the kernels' pointer tables must not be
    used any more (the products run on the torch.func path in float64 like the reference's)."""
    torch.manual_seed(0)
    model = build_mlp([16, 24, 4], ["tanh", "identity"], [True, True]).to(dev)
    params = dict(model.named_parameters())
    X, y = torch.rand(6, 16, device=dev), torch.rand(6, 4, device=dev)
    op = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)])
    v = torch.rand(op.shape[1], device=dev)
    before = op @ v
    assert op.uses_native_kernels
    model.double()
    X.data, y.data = X.data.double(), y.data.double()
    out = op @ v.double()
    assert not op.uses_native_kernels and out.dtype == torch.float64
    assert rel_err(out, before.cpu().numpy()) < 1e-5
    model.float()
    X.data, y.data = X.data.float(), y.data.float()
    assert rel_err(op @ v, before.cpu().numpy()) < 1e-6 and op.uses_native_kernels


def test_jacobian_operators_follow_swapped_parameter_storage(dev):
    """``JacobianLinearOperator`` / its transpose on the native kernels after ``p.data = t`` and a replaced dict entry."""
    torch.manual_seed(0)
    model = build_mlp([16, 24, 4], ["tanh", "identity"], [True, True]).to(dev)
    params = dict(model.named_parameters())
    data = [(torch.rand(6, 16, device=dev), torch.rand(6, 4, device=dev))]
    J = C.JacobianLinearOperator(model, params, data)
    JT = C.TransposedJacobianLinearOperator(model, params, data)
    v, u = torch.rand(J.shape[1], device=dev), torch.rand(J.shape[0], device=dev)
    for p in params.values():
        p.data = (p.data * 0.7 + 0.05).clone()
    assert rel_err(J @ v, (C.JacobianLinearOperator(model, params, data) @ v).cpu().numpy()) < 1e-6
    assert rel_err(JT @ u, (C.TransposedJacobianLinearOperator(model, params, data) @ u).cpu().numpy()) < 1e-6


def test_grouped_kronecker_blocks_see_inplace_factor_updates(dev):
    """A block-diagonal operator reads its blocks' LIVE factors on every product (reference ``block_diagonal.py`` loops
    over the blocks): in-place updates of a factor -- also ``.data`` ones, which no version counter sees -- show up.
    Equal-shape blocks run as one batched product that references the factors where they lie (``clo_gemm_ptrs_f32``)."""
    torch.manual_seed(0)
    blocks = []
    for _ in range(4):
        A, B = torch.rand(6, 6, device=dev), torch.rand(5, 5, device=dev)
        blocks.append(C.KroneckerProductLinearOperator(A + A.T, B + B.T))
    bd = C.BlockDiagonalLinearOperator(blocks)
    x = torch.rand(bd.shape[1], device=dev)
    dense = lambda: torch.block_diag(*[torch.kron(b[0], b[1]) for b in blocks])  # noqa: E731
    _ = bd @ x
    with torch.no_grad():
        blocks[2][0].mul_(0.5).add_(torch.eye(6, device=dev))
    assert rel_err(bd @ x, (dense() @ x).cpu().numpy()) < TOL
    blocks[1][1].data.mul_(1.7)
    blocks[3][0].data.copy_(blocks[0][0].data * 0.3)
    assert rel_err(bd @ x, (dense() @ x).cpu().numpy()) < TOL
    blocks[0][0].data.mul_(2.0)
    assert rel_err(bd @ x, (dense() @ x).cpu().numpy()) < TOL


def test_fuzz_kfac_operators_gpu():
    """30 random small nets (conv / linear stacks, losses, Fisher types, expand / reduce, joint / separate bias,
    input scales 1e-2 ... 1e2): KFAC / EKFAC products and damped inverses in float32 on the GPU against this
    package's float64 CPU path (pinned to the reference by the goldens) -- tools/fuzz_kfac.py."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_kfac

    worst, failures = fuzz_kfac.run(seed=3, ncase=60)
    assert not failures, "\\n".join(failures)
    assert worst < 5e-3
