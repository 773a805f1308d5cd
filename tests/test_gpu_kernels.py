"""GPU parity tests of the raw C-ABI kernels (through the ctypes binding) against float64 CPU
references.  Tolerance: max|y - y_ref| / max|y_ref| <= 1e-4 for fp32 device
results; the kernels here are expected to do ~10x better, which the asserts enforce."""

import numpy as np
import pytest
import torch

from conftest import load_golden, mlp_case_tensors
from oracle import mlp_numpy as O

pytestmark = pytest.mark.gpu

TOL = 2e-5


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


@pytest.fixture(scope="module")
def hip():
    from curvlinops_amd import _hip

    _hip.load()
    return _hip


def dev(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()


# ------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [
    (128, 128, 64), (256, 384, 128), (1, 1, 1), (7, 5, 3), (130, 257, 33), (64, 2689, 2688),
    (2688, 10, 512), (10, 300, 17), (513, 129, 1000),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(hip, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.rand((K, M) if ta else (M, K), generator=g, dtype=torch.float64) - 0.5
    B = torch.rand((N, K) if tb else (K, N), generator=g, dtype=torch.float64) - 0.5
    Ad, Bd = A.float().cuda(), B.float().cuda()
    Av = Ad.T if ta else Ad
    Bv = Bd.T if tb else Bd
    out = hip.gemm(Av, Bv)
    ref = (A.T if ta else A) @ (B.T if tb else B)
    assert rel_err(out.cpu(), ref) < TOL


def test_gemm_alpha_beta_and_splitk(hip):
    g = torch.Generator().manual_seed(0)
    A = torch.rand(100, 3000, generator=g, dtype=torch.float64) - 0.5
    B = torch.rand(3000, 70, generator=g, dtype=torch.float64) - 0.5
    C = torch.rand(100, 70, generator=g, dtype=torch.float64)
    for splitk in (1, 4, 13):
        out = C.float().cuda()
        hip.gemm(A.float().cuda(), B.float().cuda(), out=out, alpha=0.5, beta=-2.0, splitk=splitk)
        assert rel_err(out.cpu(), 0.5 * A @ B - 2.0 * C) < TOL


def test_gemm_batched_and_broadcast(hip):
    g = torch.Generator().manual_seed(1)
    A = torch.rand(5, 33, 47, generator=g, dtype=torch.float64) - 0.5
    B = torch.rand(47, 29, generator=g, dtype=torch.float64) - 0.5
    out = hip.gemm(A.float().cuda(), B.float().cuda())
    assert rel_err(out.cpu(), A @ B) < TOL
    B3 = torch.rand(5, 47, 29, generator=g, dtype=torch.float64) - 0.5
    out = hip.gemm(A.float().cuda(), B3.float().cuda())
    assert rel_err(out.cpu(), A @ B3) < TOL


# LDS-DMA engine (csrc/gemm_v3.hip; 128 x 128 x 32 tiles): one tile per workgroup, split-K slabs, stream-K (-1), with
# ragged M / N / K, every operand layout, batches, alpha / beta
@pytest.mark.parametrize("M,N,K", [(512, 2304, 1152), (516, 1156, 1000), (132, 128, 4096), (2048, 2048, 68),
                                    (640, 640, 36), (300, 3000, 260), (4608, 512, 772),
                                    (256, 4608, 1160), (1024, 2304, 580), (4096, 4224, 2052)])  # teams of 2 / 8, tall tiles
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("splitk", [None, 1, 3, -1])
def test_gemm_lds_dma_engine(hip, M, N, K, ta, tb, splitk):
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K)
    A = torch.rand((K, M) if ta else (M, K), generator=g, device="cuda") - 0.5
    B = torch.rand((N, K) if tb else (K, N), generator=g, device="cuda") - 0.5
    C0 = torch.rand(M, N, generator=g, device="cuda")
    Av, Bv = (A.T if ta else A), (B.T if tb else B)
    for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
        out = C0.clone()
        hip.gemm(Av, Bv, out=out, alpha=alpha, beta=beta, splitk=splitk)
        ref = alpha * (Av.double() @ Bv.double()) + beta * C0.double()
        assert rel_err(out.cpu(), ref.cpu()) < TOL


def test_gemm_stream_k_batched_and_repeatable(hip):
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.rand(3, 384, 520, generator=g, device="cuda") - 0.5
    B = torch.rand(3, 520, 640, generator=g, device="cuda") - 0.5
    out = hip.gemm(A, B, splitk=-1)
    assert rel_err(out.cpu(), (A.double() @ B.double()).cpu()) < TOL
    # the partial accumulators of a tile are added in a fixed order: bitwise the same result every time
    A2 = torch.rand(512, 2304, generator=g, device="cuda") - 0.5
    B2 = torch.rand(2304, 1152, generator=g, device="cuda") - 0.5
    r0 = hip.gemm(A2, B2, splitk=-1)
    assert all(torch.equal(r0, hip.gemm(A2, B2, splitk=-1)) for _ in range(10))
    # two streams at once: each has its own flags
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(4):
        with torch.cuda.stream(s1):
            outs.append(hip.gemm(A2, B2, splitk=-1))
        with torch.cuda.stream(s2):
            outs.append(hip.gemm(A2, B2, splitk=-1))
    torch.cuda.synchronize()
    assert all(torch.equal(r0, o) for o in outs)


def test_gemm_unaligned_views(hip):
    g = torch.Generator().manual_seed(2)
    big = torch.rand(301, 203, generator=g, dtype=torch.float64) - 0.5
    bigd = big.float().cuda()
    A, Ad = big[1:200, 3:150], bigd[1:200, 3:150]       # odd offsets -> scalar load path
    B, Bd = big[5:152, 7:90], bigd[5:152, 7:90]
    out = hip.gemm(Ad, Bd)
    assert rel_err(out.cpu(), A @ B) < TOL


@pytest.mark.parametrize("rows,d", [(1, 1), (37, 5), (1000, 26), (4096, 151), (300, 401), (64, 2689), (50000, 27),
                                     (1000, 64), (3000, 576), (70, 1152)])
@pytest.mark.parametrize("ones", [False, True])
def test_syrk_accum(hip, rows, d, ones):
    g = torch.Generator().manual_seed(rows + d)
    X = torch.rand(rows, d, generator=g, dtype=torch.float64) - 0.3
    Xa = torch.cat([X, torch.ones(rows, 1, dtype=torch.float64)], 1) if ones else X
    dd = Xa.shape[1]
    C0 = torch.rand(dd, dd, generator=g, dtype=torch.float64)
    C0 = C0 + C0.T
    for beta in (0.0, 1.0):
        C = C0.float().cuda()
        hip.syrk_accum(C, X.float().cuda(), alpha=0.25, beta=beta, ones_col=ones)
        ref = 0.25 * Xa.T @ Xa + beta * C0
        assert rel_err(C.cpu(), ref) < TOL
        assert torch.equal(C, C.T)


@pytest.mark.parametrize("rows,d,ones", [(100_000, 6, False), (100_003, 6, True), (5000, 33, True), (20_000, 100, False),
                                         (9000, 127, True), (9000, 128, False), (4096, 16, False), (700, 15, True),
                                         (300_000, 32, False), (12_345, 64, True)])
def test_gram_tall_kernel(hip, rows, d, ones):
    """The streaming tall-skinny Gram kernel (every padded width 16 / 32 / 64 / 96 / 128, vector and
    scalar loaders, ragged last chunk, strided rows, accumulation) against float64."""
    g = torch.Generator().manual_seed(rows + d)
    Xbig = torch.rand(rows, d + 4, generator=g, dtype=torch.float64) - 0.3
    for X in (Xbig[:, :d].contiguous(), Xbig[:, :d]):          # contiguous and strided rows
        Xa = torch.cat([X, torch.ones(rows, 1, dtype=torch.float64)], 1) if ones else X
        dd = Xa.shape[1]
        C0 = torch.rand(dd, dd, generator=g, dtype=torch.float64)
        C0 = C0 + C0.T
        for beta in (0.0, 1.0):
            C = C0.float().cuda()
            hip.syrk_accum(C, X.float().cuda() if X.is_contiguous() else Xbig.float().cuda()[:, :d], alpha=0.5,
                           beta=beta, ones_col=ones, force_gram_tall=True)
            ref = 0.5 * Xa.T @ Xa + beta * C0
            assert rel_err(C.cpu(), ref) < TOL
            assert torch.equal(C, C.T)


def test_syrk_splitk_and_strided_rows(hip):
    g = torch.Generator().manual_seed(5)
    big = torch.rand(9000, 40, generator=g, dtype=torch.float64)
    X = big[:, :36]
    C = torch.zeros(36, 36).cuda()
    hip.syrk_accum(C, big.float().cuda()[:, :36], alpha=1.0, beta=0.0, splitk=7)
    assert rel_err(C.cpu(), X.T @ X) < TOL


# ------------------------------------------------------------------------ streaming helpers
def test_axpby_transpose_rowscale(hip):
    g = torch.Generator().manual_seed(3)
    for n in (1, 5, 1024, 100003):
        x, y = torch.rand(n, generator=g), torch.rand(n, generator=g)
        out = hip.axpby(y.cuda().clone(), x.cuda(), 0.3, -1.5)
        assert torch.allclose(out.cpu(), 0.3 * x - 1.5 * y, atol=1e-6)
        out = hip.axpby(torch.full((n,), float("nan")).cuda(), x.cuda(), 2.0, 0.0)
        assert torch.allclose(out.cpu(), 2.0 * x, atol=1e-6)
    for r, c in ((1, 1), (3, 700), (257, 65), (1000, 32)):
        x = torch.rand(r, c, generator=g)
        assert torch.equal(hip.transpose(x.cuda()).cpu(), x.T.contiguous())
    x, s = torch.rand(77, 5, generator=g), torch.rand(77, generator=g) + 0.1
    assert torch.allclose(hip.rowscale(x.cuda(), s.cuda()).cpu(), s[:, None] * x, atol=1e-6)
    assert torch.allclose(hip.rowscale(x.cuda(), s.cuda(), True, 0.5).cpu(), x / (s[:, None] + 0.5), atol=1e-6)


@pytest.mark.parametrize("rows,cols,K", [(1, 1, 2), (10, 513, 3), (64, 577, 8), (512, 4608, 5), (7, 3, 130), (129, 64, 64),
                                         (33, 100, 65)])
@pytest.mark.parametrize("bias", [True, False])
def test_canonical_pack_unpack(hip, rows, cols, K, bias):
    """clo_canonical_pack_f32 / _unpack_f32 against the reference's formulation (`kfac_utils.py:280-306` cat of the
    bias as the last column, `:338-385` slicing) followed by the K-major transpose: exact (pure data movement)."""
    g = torch.Generator().manual_seed(rows * 131 + cols * 7 + K)
    w = torch.rand(rows * cols, K, generator=g).cuda()
    b = torch.rand(rows, K, generator=g).cuda() if bias else None
    joint = w.view(rows, cols, K) if b is None else torch.cat([w.view(rows, cols, K), b.unsqueeze(1)], dim=1)
    ref = joint.reshape(-1, K).T.contiguous()                # [K, rows * (cols + 1)]
    got = hip.canonical_pack(w, b, rows, cols)
    assert torch.equal(got, ref)
    w2, b2 = hip.canonical_unpack(got, rows, cols, bias)
    assert torch.equal(w2, w) and (b is None or torch.equal(b2, b))


def test_kmajor_layout_travels_through_the_kfac_chain(hip):
    """K > 1 columns through P K P^T: the canonical converters hand the blocks a K-major operand (fused pack), the
    blocks answer K-major (no transposes), the result equals the per-column products and the cat / slice route."""
    import curvlinops_amd as C
    from curvlinops_amd.canonical import ToCanonicalLinearOperator, is_kmajor

    torch.manual_seed(0)
    shapes = {"0.weight": torch.Size((12, 5, 3, 3)), "0.bias": torch.Size((12,)), "1.weight": torch.Size((7, 30)),
              "2.weight": torch.Size((9, 7)), "2.bias": torch.Size((9,))}
    groups = [{"W": "0.weight", "b": "0.bias"}, {"W": "1.weight"}, {"W": "2.weight", "b": "2.bias"}]
    PT = ToCanonicalLinearOperator(shapes, groups, torch.device("cuda:0"), torch.float32)
    P = PT.adjoint()
    K = 6
    M = [torch.rand(*s, K).cuda() for s in shapes.values()]
    canon = PT._matmat(M)
    assert is_kmajor(canon[0]) and is_kmajor(canon[2]) and not is_kmajor(canon[1])
    ref = PT._matmat([m.cpu() for m in M])
    for a, r in zip(canon, ref):
        assert torch.equal(a.cpu(), r)
    back = P._matmat(canon)
    for a, m in zip(back, M):
        assert a.is_contiguous() and torch.equal(a, m)
    # blocks: Kronecker and eigendecomposed, K-major in -> K-major out, equal to the K-trailing route
    blocks = []
    for (d_out, d_in) in ((12, 46), (7, 30), (9, 8)):
        S1, S2 = torch.rand(d_out, d_out).cuda(), torch.rand(d_in, d_in).cuda()
        blocks.append(C.KroneckerProductLinearOperator(S1 + S1.T, S2 + S2.T))
    blocks[2] = C.EighDecomposedLinearOperator(torch.rand(72).cuda(), blocks[2])
    B = C.BlockDiagonalLinearOperator(blocks)
    y_fused = B._matmat(canon)
    assert is_kmajor(y_fused[0]) and is_kmajor(y_fused[2])
    y_plain = B._matmat([c.contiguous() for c in canon])
    for a, r in zip(y_fused, y_plain):
        assert rel_err(a.cpu().numpy(), r.cpu().numpy()) < 1e-6
    out = P._matmat(y_fused)
    out_ref = P._matmat(y_plain)
    for a, r in zip(out, out_ref):
        assert a.shape == r.shape and rel_err(a.cpu().numpy(), r.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("n", [0, 1, 3, 1000, 4097, 1 << 20, (1 << 22) + 5])
def test_dot_kernel(hip, n):
    g = torch.Generator().manual_seed(n)
    x, y = torch.rand(n + 1, generator=g) - 0.5, torch.rand(n + 1, generator=g) - 0.5
    ref = float(x[:n].double() @ y[:n].double())
    got = float(hip.dot(x[:n].cuda(), y[:n].cuda(), scale=0.5))
    assert abs(got - 0.5 * ref) <= 1e-5 * max(1.0, abs(ref))
    if n:  # unaligned views take the scalar path
        xu, yu = x.cuda()[1:], y.cuda()[1:]
        assert abs(float(hip.dot(xu, yu)) - float(x[1:].double() @ y[1:].double())) <= 1e-5 * max(1.0, abs(ref))


def test_pack_probes(hip):
    P = hip.pack_probes(100001, 3, 1234, "rademacher", "cuda")
    assert P.shape == (100001, 3)
    assert torch.all(P.abs() == 1.0)
    assert abs(P.mean().item()) < 0.01
    assert torch.equal(P, hip.pack_probes(100001, 3, 1234, "rademacher", "cuda"))
    assert not torch.equal(P, hip.pack_probes(100001, 3, 1235, "rademacher", "cuda"))
    G = hip.pack_probes(200000, 2, 7, "normal", "cuda")
    assert abs(G.mean().item()) < 0.01 and abs(G.std().item() - 1.0) < 0.01
    # columns must be (nearly) uncorrelated
    assert abs((G[:, 0] * G[:, 1]).mean().item()) < 0.01
    with pytest.raises(ValueError):
        hip.pack_probes(10, 1, 0, "cauchy", "cuda")


# --------------------------------------------------------------------- MLP layer kernels
ACT_CODE = {"identity": 0, "relu": 1, "tanh": 2, "sigmoid": 3}


@pytest.mark.parametrize("N", [1, 3, 8, 13, 16])
@pytest.mark.parametrize("d_in,d_out", [(16, 8), (1024, 2688), (37, 10), (2688, 10), (260, 9)])
@pytest.mark.parametrize("act", ["relu", "tanh", "identity", "sigmoid"])
def test_fwd_jvp_layer(hip, N, d_in, d_out, act):
    g = np.random.default_rng(N * 100 + d_in + d_out)
    W = (g.random((d_out, d_in)) - 0.5) / np.sqrt(d_in)
    VW = g.random((d_out, d_in)) - 0.5
    b, Vb = g.random(d_out) - 0.5, g.random(d_out) - 0.5
    a, da = g.random((N, d_in)), g.random((N, d_in)) - 0.5
    z = a @ W.T + b
    dz = da @ W.T + a @ VW.T + Vb
    out, p1, _ = O._act(act, z)
    a_o, da_o, dphi = hip.mlp_fwd_jvp_layer(dev(W), dev(b), dev(VW), dev(Vb), dev(a), dev(da), ACT_CODE[act])
    assert rel_err(a_o.cpu(), out) < TOL
    assert rel_err(da_o.cpu(), p1 * dz) < TOL
    # first-layer form (no incoming tangent, no bias) and pure forward
    a_o, da_o, _ = hip.mlp_fwd_jvp_layer(dev(W), None, dev(VW), None, dev(a), None, ACT_CODE[act])
    z0 = a @ W.T
    out0, p10, _ = O._act(act, z0)
    assert rel_err(a_o.cpu(), out0) < TOL
    assert rel_err(da_o.cpu(), p10 * (a @ VW.T)) < TOL
    a_o, da_o, _ = hip.mlp_fwd_jvp_layer(dev(W), dev(b), None, None, dev(a), None, ACT_CODE[act])
    assert da_o is None and rel_err(a_o.cpu(), out) < TOL


@pytest.mark.parametrize("kind,loss", [(0, "mse"), (1, "ce"), (2, "bce")])
@pytest.mark.parametrize("N,C", [(1, 1), (8, 10), (5, 1000), (3, 7)])
def test_loss_hessian(hip, kind, loss, N, C):
    g = np.random.default_rng(N + C)
    f, u = 3 * (g.random((N, C)) - 0.5), g.random((N, C)) - 0.5
    y = g.integers(0, C, N) if loss == "ce" else g.random((N, C))
    ref = O.loss_hessian_apply(loss, "sum", f, y, u)
    scale = 2.0 if loss == "mse" else 1.0
    w = hip.loss_hessian_apply(kind, dev(f), dev(u), 0.7 * scale)
    assert rel_err(w.cpu(), 0.7 * ref) < TOL


def test_loss_hessian_rank(hip):
    g = np.random.default_rng(0)
    N, M, C = 6, 3, 11
    f, u, aux = g.random((N, C)), g.random((N, C)) - 0.5, g.random((N, M, C)) - 0.5
    ref = 0.3 * np.einsum("nmc,nm->nc", aux, np.einsum("nmc,nc->nm", aux, u))
    w = hip.loss_hessian_apply(3, dev(f), dev(u), 0.3, aux=dev(aux))
    assert rel_err(w.cpu(), ref) < TOL


@pytest.mark.parametrize("N", [1, 8, 11, 16])
@pytest.mark.parametrize("d_in,d_out", [(16, 8), (2688, 2688), (1024, 2688), (2688, 10), (37, 10), (261, 70)])
def test_bwd_layer(hip, N, d_in, d_out):
    g = np.random.default_rng(N + d_in * 3 + d_out)
    W = (g.random((d_out, d_in)) - 0.5) / np.sqrt(d_out)
    delta, a_prev = g.random((N, d_out)) - 0.5, g.random((N, d_in))
    dphi_prev = g.random((N, d_in))
    oW0, ob0 = g.random((d_out, d_in)), g.random(d_out)
    for beta in (0.0, 1.0):
        oW, ob = dev(oW0), dev(ob0)
        dprev = hip.mlp_bwd_layer(dev(W), dev(delta), dev(a_prev), dev(dphi_prev), oW, ob, 0.5, beta, True)
        assert rel_err(oW.cpu(), 0.5 * delta.T @ a_prev + beta * oW0) < TOL
        assert rel_err(ob.cpu(), 0.5 * delta.sum(0) + beta * ob0) < TOL
        assert rel_err(dprev.cpu(), (delta @ W) * dphi_prev) < TOL


# ------------------------------------------------------------- whole-network GGN matvec
def _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, loss_kind, scale, alpha, beta, out0=None, aux=None,
                    flags=0):
    plan = hip.MLPPlan(dims, [ACT_CODE[a] for a in acts])
    plan.flags = flags  # CLO_MLP_* kernel choice (0: the library picks, 1: keep the launch chain)
    dW, db = [dev(W) for W in Ws], [None if b is None else dev(b) for b in bs]
    dVW, dVb = [dev(v) for v in vWs], [None if v is None else dev(v) for v in vbs]
    if out0 is None:
        oW = [torch.full_like(w, float("nan")) for w in dW]
        ob = [None if b is None else torch.full_like(b, float("nan")) for b in db]
    else:
        oW = [dev(w) for w in out0[0]]
        ob = [None if b is None else dev(b) for b in out0[1]]
    plan.ggn_matvec(dW, db, dVW, dVb, oW, ob, dev(X), loss_kind, scale, alpha, beta, aux=aux)
    torch.cuda.synchronize()
    return [w.cpu().numpy() for w in oW], [None if b is None else b.cpu().numpy() for b in ob]


LOSS_KIND = {"mse": 0, "ce": 1, "bce": 2}


@pytest.mark.parametrize("case", sorted(load_golden("mlp_curvature")))
def test_ggn_matvec_vs_oracle_per_batch(hip, golden_mlp, case):
    rec = golden_mlp[case]
    dims, acts, bias, loss, red, Ws, bs, data = mlp_case_tensors(rec)
    vWs, vbs = O.unflatten_params(rec["v"], [W.shape for W in Ws], bias)
    for X, y in data:
        N, C = X.shape[0], dims[-1]
        rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, red, vWs, vbs)
        c = O.reduction_factor(loss, red, N, C)
        scale = (2.0 if loss == "mse" else 1.0) * c
        gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, LOSS_KIND[loss], scale, 1.0, 0.0)
        ref = O.flatten_params(rW, rb)
        got = O.flatten_params(gW, gb)
        assert rel_err(got, ref) < 1e-4, case


@pytest.mark.parametrize("N", [8, 16, 24, 200])
def test_ggn_matvec_large_layers(hip, N):
    """C2-like shapes (scaled down in width so the float64 oracle stays fast)."""
    g = np.random.default_rng(N)
    dims, acts = [256, 672, 672, 10], ["relu", "relu", "identity"]
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [g.random(dims[i + 1]) - 0.5 for i in range(3)]
    vWs = [g.random(W.shape) - 0.5 for W in Ws]
    vbs = [g.random(b.shape) - 0.5 for b in bs]
    X, y = g.random((N, dims[0])), g.random((N, dims[-1]))
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, "mse", "mean", vWs, vbs)
    scale = 2.0 / (N * dims[-1])
    out0 = ([g.random(W.shape) for W in Ws], [g.random(b.shape) for b in bs])
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, 0, scale, 0.5, 1.0, out0=out0)
    ref = O.flatten_params([0.5 * r + o for r, o in zip(rW, out0[0])], [0.5 * r + o for r, o in zip(rb, out0[1])])
    assert rel_err(O.flatten_params(gW, gb), ref) < 1e-4


@pytest.mark.parametrize("bias", [True, False])
@pytest.mark.parametrize("N,C", [(65, 1), (65, 16), (97, 10), (128, 16), (191, 7), (192, 16), (193, 16), (256, 3)])
def test_ggn_matvec_rows_path_narrow_heads(hip, N, C, bias):
    """65 ... 256 rows with a narrow last layer: everything behind delta_L in one launch up to 192 rows (head_rows_back_kernel:
    out_W_L, out_b_L, delta_{L-1}), the four-launch route beyond; head widths 1 ... 16, with and without biases,
    accumulation into a filled output (beta = 1)."""
    g = np.random.default_rng(1000 * N + C)
    dims, acts = [128, 328, 200, C], ["tanh", "relu", "identity"]
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [g.random(dims[i + 1]) - 0.5 if bias else None for i in range(3)]
    vWs = [g.random(W.shape) - 0.5 for W in Ws]
    vbs = [g.random(b.shape) - 0.5 if bias else None for b in bs]
    loss = "ce" if C > 1 else "mse"
    X = g.random((N, dims[0]))
    y = g.integers(0, C, N) if loss == "ce" else g.random((N, C))
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", vWs, vbs)
    scale = (2.0 if loss == "mse" else 1.0) * O.reduction_factor(loss, "mean", N, C)
    out0 = ([g.random(W.shape) for W in Ws], [g.random(b.shape) if bias else None for b in bs])
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, LOSS_KIND[loss], scale, 0.5, 1.0, out0=out0)
    for k, (got, r, o) in enumerate(zip(gW + (gb if bias else []), rW + (rb if bias else []), out0[0] + (out0[1] if bias else []))):
        assert rel_err(got, 0.5 * r + o) < 1e-4, f"block {k}"


@pytest.mark.parametrize("C", [1, 16])
@pytest.mark.parametrize("N,K", [(8, 4), (5, 64), (19, 16)])
def test_ggn_matmat_columns_narrowest_and_widest_head(hip, C, N, K):
    """The two-launch tangent of a narrow last layer (klast_partial / klast_finish) and the elementwise delta below it at
    the ends of their range: 1 and 16 outputs, K = 4 and 64 columns (one and two row-slot layouts), two row blocks."""
    g = np.random.default_rng(100 * C + N + K)
    dims, acts = [64, 136, 72, C], ["tanh", "sigmoid", "identity"]
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [g.random(dims[i + 1]) - 0.5 for i in range(3)]
    VWk = [g.random((*W.shape, K)) - 0.5 for W in Ws]
    Vbk = [g.random((*b.shape, K)) - 0.5 for b in bs]
    loss = "ce" if C > 1 else "mse"
    X = g.random((N, dims[0]))
    y = g.integers(0, C, N) if loss == "ce" else g.random((N, C))
    scale = (2.0 if loss == "mse" else 1.0) * O.reduction_factor(loss, "mean", N, C)
    gW, gb = _run_ggn_native_cols(hip, dims, acts, Ws, bs, X, VWk, Vbk, LOSS_KIND[loss], scale, 1.0, 0.0)
    for k in (0, K - 1):
        rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", [v[..., k] for v in VWk], [v[..., k] for v in Vbk])
        got = O.flatten_params([w[..., k] for w in gW], [b[..., k] for b in gb])
        assert rel_err(got, O.flatten_params(rW, rb)) < 1e-4, k


@pytest.mark.parametrize("loss", ["mse", "ce", "bce"])
@pytest.mark.parametrize("N", [9, 12, 16, 17, 25, 32, 33, 40, 48, 49, 57, 64])
def test_ggn_matvec_mid_rows_chain(hip, N, loss):
    """9 ... 64 rows: the MFMA streaming chain (mid_fwd / head / mid_dprev / mid_outer kernels) against
    the float64 oracle on a 4-layer net (two finished hidden layers, slab ping-pong in the data chain,
    a layer without bias), plain and accumulating (beta = 1) products."""
    g = np.random.default_rng(100 * N + len(loss))
    dims, acts = [64, 96, 80, 48, 7], ["tanh", "relu", "sigmoid", "identity"]
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(4)]
    bs = [g.random(dims[1]) - 0.5, None, g.random(dims[3]) - 0.5, g.random(dims[4]) - 0.5]
    vWs = [g.random(W.shape) - 0.5 for W in Ws]
    vbs = [None if b is None else g.random(b.shape) - 0.5 for b in bs]
    X = g.random((N, dims[0]))
    y = g.integers(0, dims[-1], N) if loss == "ce" else (g.integers(0, 2, (N, dims[-1])).astype(float) if loss == "bce"
                                                         else g.random((N, dims[-1])))
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", vWs, vbs)
    c = O.reduction_factor(loss, "mean", N, dims[-1])
    scale = (2.0 if loss == "mse" else 1.0) * c
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, LOSS_KIND[loss], scale, 1.0, 0.0)
    assert rel_err(O.flatten_params(gW, gb), O.flatten_params(rW, rb)) < 1e-4
    out0 = ([g.random(W.shape) for W in Ws], [None if b is None else g.random(b.shape) for b in bs])
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, LOSS_KIND[loss], scale, 0.5, 1.0, out0=out0)
    ref = O.flatten_params([0.5 * r + o for r, o in zip(rW, out0[0])],
                           [None if r is None else 0.5 * r + o for r, o in zip(rb, out0[1])])
    assert rel_err(O.flatten_params(gW, gb), ref) < 1e-4


def _mega_case(g, dims, acts, N, loss, bias=(True, True, True)):
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [g.random(dims[i + 1]) - 0.5 if bias[i] else None for i in range(3)]
    vWs = [g.random(W.shape) - 0.5 for W in Ws]
    vbs = [None if b is None else g.random(b.shape) - 0.5 for b in bs]
    X = g.random((N, dims[0]))
    C = dims[-1]
    y = g.integers(0, C, N) if loss == "ce" else (g.integers(0, 2, (N, C)).astype(float) if loss == "bce"
                                                  else g.random((N, C)))
    return Ws, bs, vWs, vbs, X, y


@pytest.mark.parametrize("loss", ["mse", "ce", "bce"])
@pytest.mark.parametrize("N", [1, 5, 8])
@pytest.mark.parametrize("dims,acts,bias", [
    ([64, 256, 512, 10], ["relu", "tanh", "identity"], (True, True, True)),          # one k-step per range, empty slices
    ([128, 272, 520, 3], ["sigmoid", "relu", "identity"], (True, False, True)),      # ragged K ranges / feature blocks
    ([16, 2816, 2688, 16], ["tanh", "relu", "identity"], (False, True, False)),      # the widest tile, C = 16
    ([1024, 1024, 1024, 1], ["relu", "identity", "identity"], (True, True, True)),   # C = 1
])
def test_ggn_matvec_persistent_kernel(hip, dims, acts, bias, N, loss):
    """Round 3: the <= 8-row matvec of a three-layer net as ONE persistent launch (mlp_mega.hip) against the
    float64 oracle, plain and accumulating, and against the six-launch chain (flags = CLO_MLP_NO_PERSISTENT)."""
    g = np.random.default_rng(7 * N + len(loss) + dims[1])
    Ws, bs, vWs, vbs, X, y = _mega_case(g, dims, acts, N, loss, bias)
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", vWs, vbs)
    scale = (2.0 if loss == "mse" else 1.0) * O.reduction_factor(loss, "mean", N, dims[-1])
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, LOSS_KIND[loss], scale, 1.0, 0.0)
    for k, (got, ref) in enumerate(zip(gW + gb, rW + rb)):  # every parameter block on its own scale
        if ref is not None:
            assert rel_err(got, ref) < 1e-4, f"block {k}"
    out0 = ([g.random(W.shape) for W in Ws], [None if b is None else g.random(b.shape) for b in bs])
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, LOSS_KIND[loss], scale, 0.5, 1.0, out0=out0)
    ref = O.flatten_params([0.5 * r + o for r, o in zip(rW, out0[0])],
                           [None if r is None else 0.5 * r + o for r, o in zip(rb, out0[1])])
    assert rel_err(O.flatten_params(gW, gb), ref) < 1e-4
    oW, ob = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, LOSS_KIND[loss], scale, 0.5, 1.0, out0=out0,
                             flags=hip.MLP_NO_PERSISTENT)
    assert rel_err(O.flatten_params(gW, gb), O.flatten_params(oW, ob)) < 2e-5


@pytest.mark.parametrize("N,loss", [(8, "mse"), (5, "ce"), (8, "bce")])
def test_ggn_matvec_persistent_kernel_c2(hip, N, loss):
    """The benchmark network itself (1024-2688-2688-10, ReLU) through the persistent kernel: float64 oracle,
    every parameter block on its own scale; 20 products on ONE workspace are bit-for-bit identical (fixed
    summation orders, the counters recycle correctly from call to call)."""
    g = np.random.default_rng(N)
    dims, acts = [1024, 2688, 2688, 10], ["relu", "relu", "identity"]
    Ws, bs, vWs, vbs, X, y = _mega_case(g, dims, acts, N, loss)
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", vWs, vbs)
    scale = (2.0 if loss == "mse" else 1.0) * O.reduction_factor(loss, "mean", N, dims[-1])
    plan = hip.MLPPlan(dims, [ACT_CODE[a] for a in acts])
    dW, db = [dev(W) for W in Ws], [dev(b) for b in bs]
    dVW, dVb = [dev(v) for v in vWs], [dev(v) for v in vbs]
    dX = dev(X)
    first = None
    for it in range(20):
        oW = [torch.full_like(w, float("nan")) for w in dW]
        ob = [torch.full_like(b, float("nan")) for b in db]
        plan.ggn_matvec(dW, db, dVW, dVb, oW, ob, dX, LOSS_KIND[loss], scale, 1.0, 0.0)
        torch.cuda.synchronize()
        got = [t.cpu().numpy() for t in oW + ob]
        if first is None:
            first = got
            for k, (a, r) in enumerate(zip(got, rW + rb)):
                assert rel_err(a, r) < 1e-4, f"block {k}"
        else:
            assert all(np.array_equal(a, b) for a, b in zip(got, first)), f"call {it} differs"


@pytest.mark.parametrize("N,loss", [(9, "mse"), (16, "ce"), (17, "bce"), (32, "mse"), (33, "ce"), (48, "bce"), (64, "mse")])
def test_ggn_matvec_mid_rows_chain_c2(hip, N, loss):
    """The benchmark network at 9 ... 64 rows (the all-MFMA chain; round 6: three products per step in its forward kernel)
    against the float64 oracle, every parameter block on its own scale, at every tile count NT = 1 ... 4 and on both sides
    of each boundary; two calls are bit-identical."""
    g = np.random.default_rng(100 + N)
    dims, acts = [1024, 2688, 2688, 10], ["relu", "relu", "identity"]
    Ws, bs, vWs, vbs, X, y = _mega_case(g, dims, acts, N, loss)
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", vWs, vbs)
    scale = (2.0 if loss == "mse" else 1.0) * O.reduction_factor(loss, "mean", N, dims[-1])
    plan = hip.MLPPlan(dims, [ACT_CODE[a] for a in acts])
    dW, db = [dev(W) for W in Ws], [dev(b) for b in bs]
    dVW, dVb = [dev(v) for v in vWs], [dev(v) for v in vbs]
    dX = dev(X)
    outs = []
    for _ in range(2):
        oW = [torch.full_like(w, float("nan")) for w in dW]
        ob = [torch.full_like(b, float("nan")) for b in db]
        plan.ggn_matvec(dW, db, dVW, dVb, oW, ob, dX, LOSS_KIND[loss], scale, 1.0, 0.0)
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in oW + ob])
    for k, (a, r) in enumerate(zip(outs[0], rW + rb)):
        assert rel_err(a, r) < 1e-4, f"block {k}"
    assert all(np.array_equal(a, b) for a, b in zip(*outs))


@pytest.mark.gpu
@pytest.mark.parametrize("N,loss", [(65, "mse"), (100, "ce"), (128, "bce"), (129, "mse"), (160, "ce")])
def test_ggn_matvec_rows_chain_c2(hip, N, loss):
    """The benchmark network at 65 ... 160 rows against the float64 oracle (round 6: up to 128 rows the outer products of the two
    hidden layers leave in ONE launch of the streaming kernel, 5 ... 8 row tiles of 16; 129 ... 192 rows run the fused forward on
    64-row tiles), accumulation into a filled output; two calls are bit-identical."""
    g = np.random.default_rng(300 + N)
    dims, acts = [1024, 2688, 2688, 10], ["relu", "relu", "identity"]
    Ws, bs, vWs, vbs, X, y = _mega_case(g, dims, acts, N, loss)
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", vWs, vbs)
    scale = (2.0 if loss == "mse" else 1.0) * O.reduction_factor(loss, "mean", N, dims[-1])
    plan = hip.MLPPlan(dims, [ACT_CODE[a] for a in acts])
    dW, db = [dev(W) for W in Ws], [dev(b) for b in bs]
    dVW, dVb = [dev(v) for v in vWs], [dev(v) for v in vbs]
    dX = dev(X)
    fill = [g.random(r.shape) * np.abs(r).mean() for r in rW + rb]   # (on the scale of the block it is added to)
    outs = []
    for _ in range(2):
        oW = [dev(f) for f in fill[:3]]
        ob = [dev(f) for f in fill[3:]]
        plan.ggn_matvec(dW, db, dVW, dVb, oW, ob, dX, LOSS_KIND[loss], scale, 1.0, 1.0)
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in oW + ob])
    for k, (a, r, f) in enumerate(zip(outs[0], rW + rb, fill)):
        assert rel_err(a, r + f) < 1e-4, f"block {k}"
    assert all(np.array_equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("side_gemm", [False, True])
def test_ggn_matvec_persistent_kernel_concurrent_streams(hip, side_gemm):
    """Round 4: the persistent kernel next to other work.  Two HIP streams each issue 50 products of the benchmark
    network (one plan = one workspace per stream; the library chains persistent launches of different streams by
    events), optionally with 4096^3 GEMMs looping on a third stream so that the 256-workgroup grid has
    to become resident while other kernels hold CUs.  Every product must equal the serial run bit for bit, and
    nothing may trap or hang."""
    g = np.random.default_rng(11)
    dims, acts, N = [1024, 2688, 2688, 10], ["relu", "relu", "identity"], 8
    Ws, bs, vWs, vbs, X, y = _mega_case(g, dims, acts, N, "mse")
    scale = 2.0 * O.reduction_factor("mse", "mean", N, dims[-1])
    dW, db, dX = [dev(W) for W in Ws], [dev(b) for b in bs], dev(X)
    nvec = 4
    dV = [([dev(g.random(W.shape) - 0.5) for W in Ws], [dev(g.random(b.shape) - 0.5) for b in bs]) for _ in range(nvec)]
    plans = [hip.MLPPlan(dims, [ACT_CODE[a] for a in acts]) for _ in range(2)]

    def outputs():
        return [torch.full_like(w, float("nan")) for w in dW], [torch.full_like(b, float("nan")) for b in db]

    serial = []
    for k in range(nvec):
        oW, ob = outputs()
        plans[0].ggn_matvec(dW, db, dV[k][0], dV[k][1], oW, ob, dX, 0, scale, 1.0, 0.0)
        torch.cuda.synchronize()
        serial.append([t.clone() for t in oW + ob])
    streams = [torch.cuda.Stream() for _ in range(3)]
    for p_, s_ in zip(plans, streams):  # workspaces (and their counters) are created on the stream that uses them
        with torch.cuda.stream(s_):
            p_.workspace(N, dX.device)
    torch.cuda.synchronize()
    A = torch.randn(4096, 4096, device="cuda") if side_gemm else None
    results = [[], []]
    for it in range(50):
        if side_gemm and it % 5 == 0:
            with torch.cuda.stream(streams[2]):
                for _ in range(4):
                    hip.gemm(A, A)
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                oW, ob = outputs()
                k = (it + i) % nvec
                plans[i].ggn_matvec(dW, db, dV[k][0], dV[k][1], oW, ob, dX, 0, scale, 1.0, 0.0)
                results[i].append((k, oW + ob))
    torch.cuda.synchronize()
    for i in range(2):
        for it, (k, got) in enumerate(results[i]):
            assert all(torch.equal(a, b) for a, b in zip(got, serial[k])), f"stream {i}, product {it} differs"


def test_ggn_matvec_persistent_kernel_rank1(hip):
    """Empirical-Fisher / MC output curvature (rank-M) through the persistent kernel."""
    g = np.random.default_rng(3)
    dims, acts, N, M = [64, 256, 512, 10], ["relu", "tanh", "identity"], 6, 3
    Ws, bs, vWs, vbs, X, _ = _mega_case(g, dims, acts, N, "mse")
    aux = g.random((N, M, dims[-1])) - 0.5
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, 3, 0.25, 1.0, 0.0, aux=dev(aux))
    oW, ob = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, 3, 0.25, 1.0, 0.0, aux=dev(aux),
                             flags=hip.MLP_NO_PERSISTENT)
    assert rel_err(O.flatten_params(gW, gb), O.flatten_params(oW, ob)) < 2e-5


@pytest.mark.parametrize("dims,acts", [([20, 36, 10], ["relu", "identity"]),            # one hidden layer, ragged widths
                                       ([32, 16, 16], ["tanh", "identity"]),            # C = 16 (widest narrow head)
                                       ([300, 520, 260, 3], ["sigmoid", "relu", "identity"])])
@pytest.mark.parametrize("N", [11, 16, 29, 37, 64])
def test_ggn_matvec_mid_rows_shapes_and_rank1(hip, dims, acts, N):
    """The 9 ... 64-row chain on ragged / minimal shapes, and with the rank-M output curvature of the
    empirical Fisher (`aux`), against the float64 oracle."""
    g = np.random.default_rng(N + dims[1])
    L = len(dims) - 1
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(L)]
    bs = [g.random(dims[i + 1]) - 0.5 for i in range(L)]
    vWs = [g.random(W.shape) - 0.5 for W in Ws]
    vbs = [g.random(b.shape) - 0.5 for b in bs]
    X, y = g.random((N, dims[0])), g.random((N, dims[-1]))
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, "mse", "sum", vWs, vbs)
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, 0, 2.0, 1.0, 0.0)
    assert rel_err(O.flatten_params(gW, gb), O.flatten_params(rW, rb)) < 1e-4
    rW, rb = O.ef_matvec_batch(Ws, bs, acts, X, y, "mse", "sum", vWs, vbs)
    f = O.forward(Ws, bs, acts, X)[0][-1]
    aux = dev(2.0 * (f - y)).reshape(N, 1, dims[-1]).contiguous()  # per-sample gradients of the sum-MSE
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, 3, 1.0, 1.0, 0.0, aux=aux)
    assert rel_err(O.flatten_params(gW, gb), O.flatten_params(rW, rb)) < 1e-4


@pytest.mark.parametrize("d_in,N", [(1024, 12), (1024, 31), (1024, 40), (260, 16), (260, 64), (400, 9)])
def test_ggn_matvec_mid_rows_first_layer_kernel(hip, d_in, N):
    """A first layer wide enough for mid_full_kernel (>= 2048 features: in-block split-K, no slabs, no finish
    launch; K ranges of 128 per wave and ragged ones), followed by two more hidden layers, against the
    float64 oracle."""
    g = np.random.default_rng(N + d_in)
    dims, acts = [d_in, 2064, 48, 32, 5], ["tanh", "relu", "sigmoid", "identity"]
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(4)]
    bs = [g.random(dims[1]) - 0.5, None, g.random(dims[3]) - 0.5, g.random(dims[4]) - 0.5]
    vWs = [g.random(W.shape) - 0.5 for W in Ws]
    vbs = [None if b is None else g.random(b.shape) - 0.5 for b in bs]
    X, y = g.random((N, dims[0])), g.random((N, dims[-1]))
    rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, "mse", "mean", vWs, vbs)
    scale = 2.0 * O.reduction_factor("mse", "mean", N, dims[-1])
    gW, gb = _run_ggn_native(hip, dims, acts, Ws, bs, X, vWs, vbs, 0, scale, 1.0, 0.0)
    assert rel_err(O.flatten_params(gW, gb), O.flatten_params(rW, rb)) < 1e-4


def _run_ggn_native_cols(hip, dims, acts, Ws, bs, X, VWk, Vbk, loss_kind, scale, alpha, beta, out0=None,
                         aux=None, pad=0):
    """K columns through clo_mlp_ggn_matmat: VWk[l] is [d_out, d_in, K], Vbk[l] is [d_out, K]; `pad`
    extra columns make the row stride ldk = K + pad differ from K."""
    plan = hip.MLPPlan(dims, [ACT_CODE[a] for a in acts])
    dW, db = [dev(W) for W in Ws], [None if b is None else dev(b) for b in bs]
    plan.bind_params(dW, db)
    K = VWk[0].shape[-1]
    ldk = K + pad

    def padded(t, fill):
        full = torch.full((*t.shape[:-1], ldk), fill, dtype=torch.float32, device="cuda")
        full[..., :K] = dev(t)
        return full

    dV = [padded(v, 7.0) for v in VWk]
    dVb = [None if v is None else padded(v, 7.0) for v in Vbk]
    if out0 is None:
        oW = [torch.full_like(v, float("nan")) for v in dV]
        ob = [None if v is None else torch.full_like(v, float("nan")) for v in dVb]
    else:
        oW = [padded(w, 0.0) for w in out0[0]]
        ob = [None if b is None else padded(b, 0.0) for b in out0[1]]
    Xd = dev(X)
    ws = plan.matmat_workspace(K, "cuda")
    ptr = lambda lst: [None if t is None else t.data_ptr() for t in lst]  # noqa: E731
    plan.ggn_matmat_ptrs(ptr(dV), ptr(dVb), ptr(oW), ptr(ob), ldk, K, Xd.data_ptr(), Xd.shape[0], loss_kind,
                         scale, alpha, beta, None if aux is None else aux.data_ptr(),
                         1 if aux is None else aux.shape[1], ws.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return ([w[..., :K].cpu().numpy() for w in oW],
            [None if b is None else b[..., :K].cpu().numpy() for b in ob])


@pytest.mark.parametrize("loss", ["mse", "ce", "bce"])
@pytest.mark.parametrize("N,K,pad", [(1, 4, 0), (5, 8, 0), (8, 32, 0), (8, 64, 0), (11, 12, 4), (3, 20, 0), (19, 16, 0)])
def test_ggn_matmat_columns_vs_oracle(hip, loss, N, K, pad):
    """K-trailing multi-column kernels (tangent-weight stream, tangent GEMMs, result stream) against the
    float64 oracle applied column by column; accumulation (beta = 1, alpha = 0.5) included."""
    g = np.random.default_rng(100 * N + K)
    dims = [64, 96, 48, 10]
    acts = ["tanh", "relu", "identity"] if loss != "bce" else ["sigmoid", "tanh", "identity"]
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [g.random(dims[i + 1]) - 0.5 for i in range(3)]
    VWk = [g.random((*W.shape, K)) - 0.5 for W in Ws]
    Vbk = [g.random((*b.shape, K)) - 0.5 for b in bs]
    X = g.random((N, dims[0]))
    y = g.integers(0, dims[-1], N) if loss == "ce" else g.random((N, dims[-1]))
    c = O.reduction_factor(loss, "mean", N, dims[-1])
    scale = (2.0 if loss == "mse" else 1.0) * c
    out0 = ([g.random(v.shape) for v in VWk], [g.random(v.shape) for v in Vbk])
    gW, gb = _run_ggn_native_cols(hip, dims, acts, Ws, bs, X, VWk, Vbk, LOSS_KIND[loss], scale, 0.5, 1.0,
                                  out0=out0, pad=pad)
    for k in range(K):
        rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", [v[..., k] for v in VWk],
                                    [v[..., k] for v in Vbk])
        ref = O.flatten_params([0.5 * r + o[..., k] for r, o in zip(rW, out0[0])],
                               [0.5 * r + o[..., k] for r, o in zip(rb, out0[1])])
        got = O.flatten_params([w[..., k] for w in gW], [b[..., k] for b in gb])
        assert rel_err(got, ref) < 1e-4, k


def test_ggn_matmat_columns_c2_width(hip):
    """C2-like widths (several K ranges per wave, 2688-wide GEMMs) at K = 32, no bias, beta = 0."""
    g = np.random.default_rng(5)
    dims, acts, N, K = [256, 672, 672, 10], ["relu", "relu", "identity"], 8, 32
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [None, None, None]
    VWk = [g.random((*W.shape, K)) - 0.5 for W in Ws]
    X, y = g.random((N, dims[0])), g.random((N, dims[-1]))
    scale = 2.0 / (N * dims[-1])
    gW, _ = _run_ggn_native_cols(hip, dims, acts, Ws, bs, X, VWk, [None] * 3, 0, scale, 1.0, 0.0)
    for k in (0, 13, 31):
        rW, _ = O.ggn_matvec_batch(Ws, bs, acts, X, y, "mse", "mean", [v[..., k] for v in VWk], [None] * 3)
        assert rel_err(np.concatenate([w[..., k].ravel() for w in gW]), np.concatenate([r.ravel() for r in rW])) < 1e-4


@pytest.mark.parametrize("loss,N,K", [("mse", 8, 8), ("ce", 11, 12)])
def test_ggn_matmat_columns_overlapped_layers(hip, loss, N, K):
    """Layers wide enough (d_out d_in >= 2^20) for the two-stream schedule: the tangent GEMM of a layer next to its weight
    stream (joined by one elementwise pass), the delta GEMM of the layer below next to the result stream; biases, two row
    blocks, accumulation."""
    g = np.random.default_rng(17 * N + K)
    dims, acts = [128, 1088, 1088, 10], ["tanh", "relu", "identity"]
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [g.random(dims[i + 1]) - 0.5 for i in range(3)]
    VWk = [g.random((*W.shape, K)) - 0.5 for W in Ws]
    Vbk = [g.random((*b.shape, K)) - 0.5 for b in bs]
    X = g.random((N, dims[0]))
    y = g.integers(0, dims[-1], N) if loss == "ce" else g.random((N, dims[-1]))
    scale = (2.0 if loss == "mse" else 1.0) * O.reduction_factor(loss, "mean", N, dims[-1])
    out0 = ([g.random(v.shape) for v in VWk], [g.random(v.shape) for v in Vbk])
    gW, gb = _run_ggn_native_cols(hip, dims, acts, Ws, bs, X, VWk, Vbk, LOSS_KIND[loss], scale, 0.5, 1.0, out0=out0)
    for k in (0, K // 2, K - 1):
        rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, loss, "mean", [v[..., k] for v in VWk], [v[..., k] for v in Vbk])
        ref = O.flatten_params([0.5 * r + o[..., k] for r, o in zip(rW, out0[0])],
                               [0.5 * r + o[..., k] for r, o in zip(rb, out0[1])])
        got = O.flatten_params([w[..., k] for w in gW], [b[..., k] for b in gb])
        assert rel_err(got, ref) < 1e-4, k


@pytest.mark.parametrize("K", [32, 64])
def test_ggn_matmat_columns_benchmark_network_vs_oracle(hip, K):
    """The benchmarked K-column point itself: 1024-2688-2688-10 (ReLU), 8 rows, K = 32 / 64 tangent columns drawn on the
    device, three of them checked block by block against the float64 oracle."""
    g = np.random.default_rng(K)
    dims, acts, N = [1024, 2688, 2688, 10], ["relu", "relu", "identity"], 8
    Ws = [(g.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i]) for i in range(3)]
    bs = [g.random(dims[i + 1]) - 0.5 for i in range(3)]
    X, y = g.random((N, dims[0])), g.random((N, dims[-1]))
    scale = 2.0 * O.reduction_factor("mse", "mean", N, dims[-1])
    gen = torch.Generator(device="cuda").manual_seed(K)
    dV = [torch.rand(*W.shape, K, device="cuda", generator=gen) - 0.5 for W in Ws]
    dVb = [torch.rand(*b.shape, K, device="cuda", generator=gen) - 0.5 for b in bs]
    oW = [torch.full_like(v, float("nan")) for v in dV]
    ob = [torch.full_like(v, float("nan")) for v in dVb]
    plan = hip.MLPPlan(dims, [ACT_CODE[a] for a in acts])
    dW, db = [dev(W) for W in Ws], [dev(b) for b in bs]
    plan.bind_params(dW, db)
    Xd = dev(X)
    ws = plan.matmat_workspace(K, "cuda")
    ptr = lambda lst: [t.data_ptr() for t in lst]  # noqa: E731
    plan.ggn_matmat_ptrs(ptr(dV), ptr(dVb), ptr(oW), ptr(ob), K, K, Xd.data_ptr(), N, 0, scale, 1.0, 0.0, None, 1,
                         ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for k in (0, K // 2 + 1, K - 1):
        vW = [v[..., k].double().cpu().numpy() for v in dV]
        vb = [v[..., k].double().cpu().numpy() for v in dVb]
        rW, rb = O.ggn_matvec_batch(Ws, bs, acts, X, y, "mse", "mean", vW, vb)
        for blk, (got, ref) in enumerate(zip([w[..., k] for w in oW] + [b[..., k] for b in ob], rW + rb)):
            assert rel_err(got.cpu().numpy(), ref) < 1e-4, (k, blk)


def test_ggn_matmat_unsupported_shapes_report(hip):
    plan = hip.MLPPlan([6, 4], [0])
    assert not plan.matmat_supported(8, 8)       # input width not a multiple of 4
    plan = hip.MLPPlan([8, 4], [0])
    assert plan.matmat_supported(8, 8) and not plan.matmat_supported(6, 6) and not plan.matmat_supported(128, 128)


# ------------------------------------------------------------------------ Cholesky inverse
@pytest.mark.parametrize("nb", [1, 7, 16, 33, 64, 65, 80, 100, 127, 128])
def test_potrf_diag_block(hip, nb):
    """One diagonal block of the blocked inverse through the C ABI (clo_potrf_diag_f32): lower Cholesky factor in place
    and the inverse of the triangular factor; up to 64 rows one wave, 65 ... 128 the four-wave kernel."""
    import ctypes

    lib = hip.load()
    g = torch.Generator().manual_seed(nb)
    B = torch.rand(nb, nb + 3, generator=g, dtype=torch.float64) - 0.5
    A64 = B @ B.T / (nb + 3) + 0.05 * torch.eye(nb, dtype=torch.float64)
    lda = nb + 5   # a view with a leading dimension of its own
    buf = torch.full((nb, lda), 7.0, device="cuda")
    buf[:, :nb] = A64.float().cuda()
    Li = torch.full((nb, nb), 3.0, device="cuda")
    status = torch.zeros(1, device="cuda", dtype=torch.int32)
    rc = lib.clo_potrf_diag_f32(ctypes.c_void_p(buf.data_ptr()), lda, nb, ctypes.c_void_p(Li.data_ptr()), nb,
                                ctypes.c_void_p(status.data_ptr()), 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and int(status.item()) == 0
    L = torch.linalg.cholesky(A64)
    got_L = torch.tril(buf[:, :nb]).double().cpu()
    assert rel_err(got_L, L) < 1e-5
    assert torch.equal(buf[:, nb:].cpu(), torch.full((nb, lda - nb), 7.0))          # nothing outside the block
    assert rel_err(Li.double().cpu(), torch.linalg.inv(L)) < 2e-4
    assert torch.equal(torch.triu(Li, 1).cpu(), torch.zeros(nb, nb))                 # exact zeros above the diagonal
    # a non-positive pivot is reported with its 1-based index
    bad = A64.clone().float().cuda().contiguous()
    k = nb // 2
    bad[k, k] = -1.0
    status.zero_()
    rc = lib.clo_potrf_diag_f32(ctypes.c_void_p(bad.data_ptr()), nb, nb, ctypes.c_void_p(Li.data_ptr()), nb,
                                ctypes.c_void_p(status.data_ptr()), 100, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and int(status.item()) == 100 + k + 1


@pytest.mark.parametrize("n", [1, 5, 64, 65, 100, 128, 130, 192, 401, 1000])
def test_cholesky_inverse(hip, n):
    g = torch.Generator().manual_seed(n)
    B = torch.rand(n, n + 3, generator=g, dtype=torch.float64) - 0.5
    A = B @ B.T / (n + 3) + 0.05 * torch.eye(n, dtype=torch.float64)
    for damping in (0.0, 1e-2):
        got = hip.cholesky_inverse(A.float().cuda(), damping)
        ref = torch.linalg.inv(A + damping * torch.eye(n, dtype=torch.float64))
        assert rel_err(got.cpu(), ref) < 1e-3
        assert torch.equal(got, got.T)


def test_cholesky_pipeline_orders_and_repeatability(hip):
    """Orders >= 1536 run the pipelined factorisation + inverse (look-ahead on the caller's stream, trailing updates and
    the triangular inverse on a helper stream; reference `kronecker.py:328-373`): block-aligned and ragged orders, single,
    batched and operator-level (worker threads) -- residual against float64 and run-to-run bit equality (a race between
    the streams would show up as differing results)."""
    from curvlinops_amd import linalg_native

    g = torch.Generator().manual_seed(11)

    def spd(n):
        X = torch.randn(n + 64, n, generator=g).cuda()
        return X.T @ X / (n + 64)

    def residual(A, X, d):
        R = (A.double() + d * torch.eye(A.shape[0], dtype=torch.float64, device=A.device)) @ X.double()
        R.diagonal().sub_(1.0)
        return float(R.abs().max())

    for n in (1536, 1537, 2052):
        A = spd(n)
        X1, X2 = hip.cholesky_inverse(A, 1e-3), hip.cholesky_inverse(A, 1e-3)
        assert torch.equal(X1, X2) and residual(A, X1, 1e-3) < 5e-4
    mats = [spd(1664) for _ in range(3)]
    outs, outs2 = [torch.empty_like(m) for m in mats], [torch.empty_like(m) for m in mats]
    status = torch.zeros(3, device="cuda", dtype=torch.int32)
    damps = [1e-3, 2e-3, 5e-4]
    hip.cholesky_inverse_batched_into(mats, damps, outs, status)
    hip.cholesky_inverse_batched_into(mats, damps, outs2, status)
    assert int(status.abs().sum()) == 0
    for A, X, X2, d in zip(mats, outs, outs2, damps):
        assert torch.equal(X, X2) and residual(A, X, d) < 5e-4
    mix = [spd(n) for n in (1600, 1600, 1153, 577, 2305, 64, 129)]
    res = []
    for _ in range(2):
        with linalg_native.concurrent_inverses():
            res.append([linalg_native.damped_cholesky_inverse(A, 1e-3) for A in mix])
    torch.cuda.synchronize()
    for A, X, X2 in zip(mix, *res):
        assert torch.equal(X, X2) and residual(A, X, 1e-3) < 5e-4


def test_cholesky_inverse_from_a_worker_thread_while_another_thread_captures(hip):
    """`clo_cholesky_inverse_f32` never synchronises the device and its bookkeeping (stream / event pools) runs in relaxed
    capture mode: a worker thread may run it -- the pipelined route, helper streams and all -- while the main thread is
    inside a GLOBAL-mode `torch.cuda.graph` capture (the package captures KFAC builds itself); the capture stays valid and
    both results are right.  (Round 5 held a process-wide mutex across a hipDeviceSynchronize in this entry point.)"""
    import threading

    lib = hip.load()
    g = torch.Generator().manual_seed(5)
    n = 1700   # >= 1536: the pipelined route with the set's own streams
    X = torch.randn(n + 64, n, generator=g).cuda()
    A = X.T @ X / (n + 64)
    ref = hip.cholesky_inverse(A, 1e-3)   # warm: helper streams, events, workspaces exist
    out = torch.empty_like(A)
    status = torch.zeros(1, device="cuda", dtype=torch.int32)
    ws = torch.empty(lib.clo_cholesky_inverse_ws_floats(n), device="cuda")
    worker_stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    err = []

    def work():
        try:   # the raw entry point on preallocated buffers: nothing but the library's own calls happens in this thread
            rc = lib.clo_cholesky_inverse_f32(A.data_ptr(), n, out.data_ptr(), n, n, 1e-3, ws.data_ptr(), status.data_ptr(),
                                              worker_stream.cuda_stream)
            if rc != 0:
                err.append(hip.load().clo_last_error())
        except Exception as e:  # noqa: BLE001
            err.append(repr(e))

    x = torch.ones(4096, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):   # capture_error_mode="global" (the default)
        y = x * 2.0
        t = threading.Thread(target=work)
        t.start()
        t.join()
        z = y + 1.0
    assert not err, err
    graph.replay()
    torch.cuda.synchronize()
    assert float(z.min()) == 3.0 and float(z.max()) == 3.0
    assert int(status.item()) == 0
    assert torch.equal(out, ref)


def test_cholesky_inverse_not_pd_raises(hip):
    A = torch.eye(70)
    A[40, 40] = -1.0
    with pytest.raises(RuntimeError):
        hip.cholesky_inverse(A.cuda())
    # the retry-in-double path of the public wrapper succeeds when damping repairs the matrix
    from curvlinops_amd import linalg_native

    A[40, 40] = 1e-12
    out = linalg_native.damped_cholesky_inverse(A.cuda(), 1e-3)
    assert torch.isfinite(out).all()


def test_concurrent_inverses_match_sequential_and_retry(hip):
    """Inverses enqueued on the stream pool == the synchronous ones; a factor whose fp32 factorisation
    fails is redone in float64 into the same output tensor (warning), without retry it raises."""
    import warnings

    from curvlinops_amd import linalg_native

    g = torch.Generator().manual_seed(3)
    mats = []
    for n in (5, 64, 130, 257, 700, 33, 1000):
        X = torch.rand(n + 7, n, generator=g, dtype=torch.float64)
        mats.append((X.T @ X / n).float().cuda())
    seq = [linalg_native.damped_cholesky_inverse(A, 1e-3) for A in mats]
    with linalg_native.concurrent_inverses():
        par = [linalg_native.damped_cholesky_inverse(A, 1e-3) for A in mats]
    for a, b in zip(seq, par):
        assert torch.equal(a, b)
    # fp32 fails (the damping vanishes next to 1.0f, the 2x2 block stays singular), float64 succeeds
    bad = torch.eye(70, dtype=torch.float64)
    bad[40, 41] = bad[41, 40] = 1.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with linalg_native.concurrent_inverses():
            outs = [linalg_native.damped_cholesky_inverse(m, 1e-10) for m in (mats[1], bad.float().cuda(), mats[2])]
    ref = torch.linalg.inv(bad + 1e-10 * torch.eye(70, dtype=torch.float64))
    assert any("double precision" in str(x.message) for x in w)
    assert rel_err(outs[1].cpu(), ref) < 1e-3
    assert rel_err(outs[0].cpu(), torch.linalg.inv(mats[1].double().cpu() + 1e-10 * torch.eye(64, dtype=torch.float64))) < 5e-2
    with pytest.raises(RuntimeError):
        with linalg_native.concurrent_inverses():
            linalg_native.damped_cholesky_inverse(bad.float().cuda(), 1e-10, retry_double_precision=False)
    # the same inside a batch large enough for the worker threads
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with linalg_native.concurrent_inverses():
            outs = [linalg_native.damped_cholesky_inverse(m, 1e-10) for m in [*mats, bad.float().cuda()]]
    assert any("double precision" in str(x.message) for x in w)
    assert rel_err(outs[-1].cpu(), ref) < 1e-3


@pytest.mark.parametrize("geom", [
    dict(B=3, C=2, H=8, W=8, k=(3, 3), s=(1, 1), p=(1, 1), d=(1, 1)),
    dict(B=2, C=3, H=9, W=7, k=(3, 2), s=(2, 1), p=(0, 1), d=(1, 2)),
    dict(B=5, C=1, H=32, W=32, k=(5, 5), s=(1, 1), p=(0, 0), d=(1, 1)),
    dict(B=4, C=64, H=8, W=8, k=(3, 3), s=(2, 2), p=(1, 1), d=(1, 1)),
])
def test_im2col(hip, geom):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(geom["B"], geom["C"], geom["H"], geom["W"], generator=g)
    ref = torch.nn.functional.unfold(x, geom["k"], dilation=geom["d"], padding=geom["p"], stride=geom["s"]).transpose(1, 2)
    got = hip.im2col(x.cuda(), geom["k"], geom["s"], geom["p"], geom["d"])
    assert torch.equal(got.cpu(), ref.contiguous())


@pytest.mark.parametrize("n", [3, 64, 130, 577, 1153])
def test_cholesky_inverse_batched_equal_sizes(hip, n):
    """Factors of equal size in ONE chain of launches (one workgroup per factor in the leaves, batched
    GEMMs), odd sizes padded internally to float4-complete rows: every member equals the
    single-factor result and the float64 inverse; a non-positive-definite member is reported by its
    own status entry without disturbing the others."""
    g = torch.Generator().manual_seed(n)
    mats, damps = [], [1e-3, 2e-3, 5e-4, 1e-3]
    for _ in range(4):
        X = torch.rand(n + 5, n, generator=g, dtype=torch.float64)
        mats.append((X.T @ X / n).float().cuda())
    bad = mats[2].clone()
    bad[n // 2, n // 2] = -5.0
    members = [mats[0], mats[1], bad, mats[3]]
    outs = [torch.empty(n, n, device="cuda") for _ in members]
    status = torch.zeros(len(members), device="cuda", dtype=torch.int32)
    hip.cholesky_inverse_batched_into(members, damps, outs, status)
    st = status.cpu().tolist()
    assert st[0] == st[1] == st[3] == 0 and 0 < st[2] <= n
    for i in (0, 1, 3):
        single = hip.cholesky_inverse(members[i], damps[i])
        assert rel_err(outs[i].cpu(), single.cpu()) < 1e-3  # other split-K choices: rounding order only
        ref = torch.linalg.inv(members[i].double().cpu() + damps[i] * torch.eye(n, dtype=torch.float64))
        assert rel_err(outs[i].cpu(), ref) < 2e-3
        assert torch.equal(outs[i], outs[i].T)


def test_gemm_and_syrk_randomised(hip):
    """Random shapes (tiny ... 700), transposed views, batches, alpha / beta, forced split-K, SYRK with
    the implicit ones column: every engine of clo_gemm_f32 (aligned tiles, 64x64x64 small tiles, v1,
    the one-launch tiny kernel) against float64 (tools/fuzz_gemm.py)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gemm

    worst, failures = fuzz_gemm.run(seed=5, ncase=150)
    assert not failures, failures
    assert worst < 2e-5


# ----------------------------------------------------------------------------- fused im2col -> SYRK
@pytest.mark.parametrize("B,C_,H,W,k,s,p,d,ones", [
    (3, 2, 8, 8, (3, 3), (1, 1), (1, 1), (1, 1), False),     # 18 features
    (5, 3, 9, 7, (3, 2), (2, 1), (0, 1), (1, 2), True),      # odd patch length + bias column, dilation
    (4, 1, 32, 32, (5, 5), (1, 1), (0, 0), (1, 1), True),    # LeNet conv1: 25 + 1
    (2, 3, 32, 32, (7, 7), (2, 2), (3, 3), (1, 1), False),   # ResNet stem: 147 (not a multiple of 4)
    (64, 64, 8, 8, (3, 3), (1, 1), (1, 1), (1, 1), False),   # ResNet layer1: 576, 4096 rows (split-K)
    (16, 128, 4, 4, (3, 3), (1, 1), (1, 1), (1, 1), False),  # 1152
    (8, 64, 8, 8, (1, 1), (2, 2), (0, 0), (1, 1), False),    # 1x1 down-sampling
])
def test_im2col_syrk_fused_matches_materialised(B, C_, H, W, k, s, p, d, ones):
    """`clo_im2col_syrk_accum_f32` (patches generated in the GEMM's tile loader) == SYRK of the
    materialised `clo_im2col_f32` patches == float64 unfold; symmetric bitwise; accumulates with beta."""
    from curvlinops_amd import _hip

    dev = torch.device("cuda:0")
    torch.manual_seed(B + C_)
    x = torch.randn(B, C_, H, W, device=dev)
    P = torch.nn.functional.unfold(x.double(), k, dilation=d, padding=p, stride=s).transpose(1, 2)
    P = P.reshape(-1, P.shape[-1])
    if ones:
        P = torch.cat([P, P.new_ones(P.shape[0], 1)], dim=1)
    ref = P.T @ P
    dd = P.shape[1]
    Cf = torch.full((dd, dd), float("nan"), device=dev)
    _hip.im2col_syrk_accum(Cf, x, k, s, p, d, alpha=0.5, beta=0.0, ones_col=ones)
    assert torch.equal(Cf, Cf.T)
    assert rel_err(Cf.cpu(), (0.5 * ref).cpu().numpy()) < 2e-5
    pm = _hip.im2col(x, k, s, p, d)
    Cm = torch.empty(dd, dd, device=dev)
    _hip.syrk_accum(Cm, pm.reshape(-1, pm.shape[-1]), alpha=0.5, beta=0.0, ones_col=ones)
    assert rel_err(Cf.cpu(), Cm.double().cpu().numpy()) < 2e-5
    _hip.im2col_syrk_accum(Cf, x, k, s, p, d, alpha=0.25, beta=1.0, ones_col=ones)
    assert rel_err(Cf.cpu(), (0.75 * ref).cpu().numpy()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,C_,H,W,k,s,p,d,ones", [
    (64, 64, 8, 8, (3, 3), (1, 1), (1, 1), (1, 1), True),     # ResNet-18 layer1: 577, pixel Gram 4096^2
    (32, 128, 4, 4, (3, 3), (1, 1), (1, 1), (1, 1), True),    # layer2
    (32, 64, 8, 8, (3, 3), (2, 2), (1, 1), (1, 1), False),    # strided first conv of a stage (8x8 -> 4x4)
    (16, 256, 2, 2, (3, 3), (1, 1), (1, 1), (1, 1), True),    # layer3: every row of the patch matrix is mostly padding
    (16, 512, 1, 1, (3, 3), (1, 1), (1, 1), (1, 1), True),    # layer4: only the centre taps are ever nonzero
    (5, 5, 6, 7, (3, 3), (1, 1), (1, 1), (1, 1), True),       # odd sizes, channel groups with a tail
    (4, 3, 7, 5, (3, 2), (1, 2), (2, 0), (2, 1), True),       # anisotropic kernel / stride / padding / dilation
    (3, 6, 5, 5, (5, 5), (1, 1), (2, 2), (1, 1), False),      # 5x5 kernel
    (2, 7, 3, 3, (3, 3), (1, 1), (0, 0), (1, 1), True),       # no padding: a single output position
    (2, 32, 8, 8, (9, 9), (1, 1), (4, 4), (1, 1), True),      # large kernel: tap table + pair list take 26 KB of the LDS
])
def test_pixel_gram_fold_matches_patch_product(B, C_, H, W, k, s, p, d, ones):
    """Input covariance of a convolution from the pixel Gram ``X^T X`` (``X = x`` as ``[B, C H W]``) folded over the taps
    (``clo_patch_fold_f32``) == float64 ``unfold``-based patch product of the reference (kfac_utils.py:78-121 +
    kfac_hooks.py:350), incl. the bias row / column; bitwise symmetric; accumulates with beta."""
    from curvlinops_amd import _hip

    dev = torch.device("cuda:0")
    torch.manual_seed(B + C_ + H)
    x = torch.randn(B, C_, H, W, device=dev)
    OH = (H + 2 * p[0] - d[0] * (k[0] - 1) - 1) // s[0] + 1
    OW = (W + 2 * p[1] - d[1] * (k[1] - 1) - 1) // s[1] + 1
    assert _hip.load().clo_patch_fold_supported(C_, H, W, k[0], k[1], OH, OW)
    P = torch.nn.functional.unfold(x.double(), k, dilation=d, padding=p, stride=s).transpose(1, 2)
    P = P.reshape(-1, P.shape[-1])
    if ones:
        P = torch.cat([P, P.new_ones(P.shape[0], 1)], dim=1)
    ref = P.T @ P
    dd = P.shape[1]
    Cf = torch.full((dd, dd), float("nan"), device=dev)
    _hip.pixel_gram_accum(Cf, x, k, s, p, d, alpha=0.5, beta=0.0, ones_col=ones)
    assert torch.equal(Cf, Cf.T)
    assert rel_err(Cf.cpu(), (0.5 * ref).cpu().numpy()) < 2e-5
    _hip.pixel_gram_accum(Cf, x, k, s, p, d, alpha=0.25, beta=1.0, ones_col=ones)
    assert torch.equal(Cf, Cf.T)
    assert rel_err(Cf.cpu(), (0.75 * ref).cpu().numpy()) < 2e-5


@pytest.mark.gpu
def test_syrk_grouped_matches_float64_and_is_repeatable():
    """`clo_syrk_grouped_f32`: many covariance products of different shapes in one launch (the gradient covariances of a
    factor build) against float64 -- one and many row chunks per tile, ragged widths, ones columns, strided rows, beta
    accumulation, zero rows; results are bitwise symmetric and bitwise repeatable (slabs are summed in chunk order)."""
    from curvlinops_amd import _hip

    g = torch.Generator().manual_seed(21)
    shapes = [(131072, 64, False), (32768, 64, False), (8192, 128, False), (2048, 256, False), (512, 512, False),
              (512, 10, False), (4096, 65, True), (300, 129, False), (70000, 17, True), (0, 8, False), (1, 3, True),
              (2048, 200, False)]
    Xs, ones = [], []
    for rows, d, o in shapes:
        X = torch.randn(rows, d + 3, generator=g).cuda()[:, :d] if d % 2 else torch.randn(rows, d, generator=g).cuda()
        Xs.append(X), ones.append(o)
    dds = [d + (1 if o else 0) for _, d, o in shapes]
    alphas = [0.5 + 0.1 * i for i in range(len(shapes))]

    def ref(X, o, alpha):
        Xd = X.double()
        if o:
            Xd = torch.cat([Xd, Xd.new_ones(Xd.shape[0], 1)], dim=1)
        return alpha * (Xd.T @ Xd)

    C1 = [torch.full((dd, dd), float("nan"), device="cuda") for dd in dds]
    _hip.syrk_grouped(C1, Xs, alphas, [0.0] * len(shapes), ones)
    C2 = [torch.full((dd, dd), float("nan"), device="cuda") for dd in dds]
    _hip.syrk_grouped(C2, Xs, alphas, [0.0] * len(shapes), ones)
    for c1, c2, X, o, a in zip(C1, C2, Xs, ones, alphas):
        want = ref(X, o, a)
        assert torch.equal(c1, c1.T) and torch.equal(c1, c2)
        scale = float(want.abs().max()) or 1.0
        assert float((c1.double() - want).abs().max()) / scale < 2e-5
    _hip.syrk_grouped(C1, Xs, [0.25 * a for a in alphas], [1.0] * len(shapes), ones)   # accumulate
    for c1, X, o, a in zip(C1, Xs, ones, alphas):
        want = 1.25 * ref(X, o, a)
        scale = float(want.abs().max()) or 1.0
        assert torch.equal(c1, c1.T) and float((c1.double() - want).abs().max()) / scale < 2e-5
    # more problems than one launch takes
    many = [torch.randn(100 + 7 * i, 5 + i, generator=g).cuda() for i in range(55)]
    Cm = [torch.empty(x.shape[1], x.shape[1], device="cuda") for x in many]
    _hip.syrk_grouped(Cm, many, [1.0] * 55, [0.0] * 55)
    for c, x in zip(Cm, many):
        assert rel_err(c.cpu(), (x.double().T @ x.double()).cpu().numpy()) < 2e-5


@pytest.mark.gpu
def test_persistent_kernel_fails_soft_when_the_gpu_is_shared():
    """The persistent <= 8-row kernel needs all its 256 workgroups resident.  While ANOTHER PROCESS holds 200 compute units
    (tools/failsoft_hog.py) a launch cannot become co-resident: its bounded waits must END the kernel (garbage results) --
    no trap, no sticky context error --, the next call reports the timeout once (CLO_EASYNC -> RuntimeError), the mode is
    disabled on the device and the launch chain serves the following products (equal to the chain's reference)."""
    import json
    import os
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    victim = subprocess.Popen([sys.executable, os.path.join(root, "tools", "failsoft_victim.py")], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    hog = None
    try:
        line = victim.stdout.readline()
        assert "victim ready" in line, line + victim.stderr.read()
        hog = subprocess.Popen([sys.executable, os.path.join(root, "tools", "failsoft_hog.py"), "10"], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True)
        assert "hog running" in hog.stdout.readline()
        time.sleep(0.5)
        victim.stdin.write("go\n")
        victim.stdin.flush()
        out, err = victim.communicate(timeout=120)
        res = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    finally:
        if hog is not None:
            hog.wait(timeout=60)
        if victim.poll() is None:
            victim.kill()
    assert victim.returncode == 0, err[-2000:]
    assert res["persistent_equals_chain_when_free"] and res["status_before"] == 0
    assert res["timed_out_launch_returned"], res          # the kernel ended on its own
    assert res["timed_out_seconds"] < 8.0, res             # ... well before the hog released the CUs
    assert res["timed_out_result_has_nan"], res            # ... its result is NaN-marked, not plausible garbage
    assert res["reported"], res                            # ... and the next call said so, once
    assert res["status_after"] & 1, res                    # persistent MLP kernel disabled on this device
    assert res["next_product_equals_chain"] and res["context_alive"], res


# ---------------------------------------------------------------------------------------------
# Householder tridiagonalisation (clo_sytrd_f32) and the eigensolver built on it
# ---------------------------------------------------------------------------------------------
def _sym_case(kind: str, n: int, dev):
    g = torch.Generator(device="cpu").manual_seed(1000 + n)
    if kind == "full":
        X = torch.randn(n + 5, n, generator=g, dtype=torch.float64)
        A = X.T @ X / (n + 5)
    elif kind == "lowrank":   # a Kronecker factor of a batch with fewer rows than features
        r = max(2, n // 5)
        X = torch.randn(r, n, generator=g, dtype=torch.float64) * torch.logspace(0, -3, n, dtype=torch.float64)
        A = X.T @ X / r
    elif kind == "indefinite":
        X = torch.randn(n, n, generator=g, dtype=torch.float64)
        A = 0.5 * (X + X.T)
    elif kind == "diagonal":
        A = torch.diag(torch.linspace(-1.0, 2.0, n, dtype=torch.float64))
    elif kind == "zero":
        A = torch.zeros(n, n, dtype=torch.float64)
    else:
        raise ValueError(kind)
    return A


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["full", "lowrank", "indefinite", "diagonal", "zero"])
@pytest.mark.parametrize("n", [3, 4, 7, 64, 66, 67, 130, 333, 1030])
def test_eigh_sytrd(hip, kind, n):
    """eigh_sytrd (clo_sytrd_f32 -> hand-written divide & conquer -> block-reflector back-transformation) against
    float64 LAPACK: eigenvalues, residual |A Q - Q diag(lam)| and orthogonality |Q^T Q - I|, all <= 1e-5 relative
    to |A| (the tolerance the fp32 rocSOLVER path of torch.linalg.eigh meets on the same matrices)."""
    from curvlinops_amd.linalg_native import eigh_sytrd

    dev = torch.device("cuda:0")
    A64 = _sym_case(kind, n, dev)
    A = A64.to(dev, torch.float32)
    lam, Q = eigh_sytrd(A)
    ref = torch.linalg.eigvalsh(A64)
    scale = max(float(A64.abs().max()), 1e-30) if kind != "zero" else 1.0
    lam64, Q64 = lam.double().cpu(), Q.double().cpu()
    assert torch.all(lam64[1:] >= lam64[:-1] - 1e-6 * scale)
    assert float((lam64 - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), scale)
    assert float((A64 @ Q64 - Q64 * lam64).abs().max()) <= 1e-5 * scale * max(1.0, n ** 0.5 / 8)
    assert float((Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max()) <= 1e-5
    # the input is not modified
    assert torch.equal(A.cpu(), A64.float())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [24, 200, 1500])
def test_sytrd_is_a_similarity_transform(hip, n):
    """The tridiagonal matrix (d, e) of clo_sytrd_f32 has the spectrum of the input, and its first entries follow
    LAPACK's ssytrd sign convention (the first reflector recomputed in float64 here; later entries are not
    forward-stable, only the spectrum is).  The rocSOLVER comparison of rounds 2-3 lives in tools/probe_sytrd.py."""
    dev = torch.device("cuda:0")
    A64 = _sym_case("indefinite", n, dev)
    ld = (n + 3) // 4 * 4

    def padded():
        P = torch.zeros(n, ld, device=dev)
        P[:, :n] = A64.float().to(dev)
        return P

    A = padded()
    D, E, tau = hip.sytrd_(A, n)
    T = torch.diag(D.double().cpu()) + torch.diag(E[: n - 1].double().cpu(), 1) + torch.diag(E[: n - 1].double().cpu(), -1)
    ref = torch.linalg.eigvalsh(A64)
    assert float((torch.linalg.eigvalsh(T) - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float(tau[: n - 2].min()) >= 0.0 and float(tau[: n - 2].max()) <= 2.0   # reflector scales
    # column 0 in float64: d_0 = a_00, e_0 = -sign(a_10) |a[1:, 0]|, tau_0 = (e_0 - a_10) / e_0
    a = A64[:, 0].double().cpu()
    e0 = -torch.sign(a[1]) * a[1:].norm()
    sc = float(A64.abs().max())
    assert abs(float(D[0]) - float(a[0])) <= 1e-6 * sc
    assert abs(float(E[0]) - float(e0)) <= 1e-5 * sc
    assert abs(float(tau[0]) - float((e0 - a[1]) / e0)) <= 1e-5


@pytest.mark.gpu
def test_eigh_takes_the_native_route(hip, monkeypatch):
    """Every fp32 GPU eigh of order 3..8184 runs on the hand-written solver (one path in the package: no
    torch.linalg.eigh / rocSOLVER call unless the verification fails)."""
    from curvlinops_amd import linalg_native as L

    dev = torch.device("cuda:0")
    A = _sym_case("lowrank", 300, dev).to(dev, torch.float32)
    calls = []
    real = L.eigh_sytrd
    monkeypatch.setattr(L, "eigh_sytrd", lambda M, mb=0: (calls.append(1), real(M, mb))[1])
    monkeypatch.setattr(torch.linalg, "eigh", lambda *a, **k: (_ for _ in ()).throw(AssertionError("vendor eigh called")))
    lam1, Q1 = L.eigh(A)
    assert calls == [1]
    ref = np.linalg.eigvalsh(A.double().cpu().numpy())
    assert float(np.abs(lam1.double().cpu().numpy() - ref).max()) <= 1e-5 * float(np.abs(ref).max())
    assert float((A @ Q1 - Q1 * lam1).abs().max()) <= 1e-5 * float(A.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("power", [-8, -6, -3, 0, 4, 7])
@pytest.mark.parametrize("mode", ["native", "many"])
def test_eigh_is_scale_invariant(hip, monkeypatch, power, mode):
    """Factors of tiny norm (gradient covariances of mean-reduced losses) and of huge norm: every GPU entry point
    normalises first (rocSOLVER's tridiagonal solver behind plain torch.linalg.eigh applies an absolute tolerance:
    fp32 matrices of norm 1e-6 come back with 30 % eigenvalue error).  Relative accuracy must not depend on the
    scale."""
    from curvlinops_amd import linalg_native as L

    dev = torch.device("cuda:0")
    n = 97
    A64 = _sym_case("indefinite", n, dev) * 10.0 ** power
    A = A64.to(dev, torch.float32)
    A64 = A.double().cpu()
    if mode == "many":
        B64 = _sym_case("lowrank", n, dev) * 10.0 ** (power - 1)
        B = B64.to(dev, torch.float32)
        (lam, Q), (lam_b, Q_b) = L.eigh_many([A, B])     # equal sizes: the stacked batched call
        refb = torch.linalg.eigvalsh(B.double().cpu())
        assert float((lam_b.double().cpu() - refb).abs().max()) <= 1e-5 * float(refb.abs().max())
    else:
        lam, Q = L.eigh(A)
    ref = torch.linalg.eigvalsh(A64)
    scale = float(A64.abs().max())
    lam64, Q64 = lam.double().cpu(), Q.double().cpu()
    assert float((lam64 - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((A64 @ Q64 - Q64 * lam64).abs().max()) <= 2e-5 * scale
    assert float((Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max()) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["random", "toeplitz", "clustered", "graded", "decoupled"])
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 130, 577, 1500])
def test_tridiagonal_divide_and_conquer(hip, kind, n):
    """eigh_native.stedc_native (leaves by implicit QL, merges with deflation / float64 secular equation /
    Gu-Eisenstat weights, one batched GEMM pair per level) against float64 LAPACK on tridiagonal matrices:
    well separated, clustered (deflation), graded and block-decoupled spectra."""
    from curvlinops_amd import eigh_native

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7 * n + len(kind))
    if kind == "random":
        d, e = torch.rand(n, generator=g) - 0.5, torch.rand(n, generator=g) - 0.5
    elif kind == "toeplitz":
        d, e = torch.full((n,), 2.0), torch.full((n,), -1.0)
    elif kind == "clustered":
        d = torch.cat([torch.ones(n // 2), torch.rand(n - n // 2, generator=g)])
        e = torch.rand(n, generator=g) * 1e-6
    elif kind == "graded":
        d = torch.logspace(0, -6, n)
        e = torch.logspace(0, -6, n) * 0.3
    else:
        d, e = torch.rand(n, generator=g), torch.rand(n, generator=g)
        e[::7] = 0.0
    T = torch.diag(d.double()) + torch.diag(e[: n - 1].double(), 1) + torch.diag(e[: n - 1].double(), -1)
    ref = torch.linalg.eigvalsh(T)
    lam, Q = eigh_native.stedc_native(d.to(dev), e.to(dev), n)
    lam64, Q64 = lam.double().cpu(), Q.double().cpu()
    scale = max(float(T.abs().max()), 1e-30)
    assert torch.all(lam64[1:] >= lam64[:-1])
    assert float((lam64 - ref).abs().max()) <= 5e-6 * scale
    assert float((T @ Q64 - Q64 * lam64).abs().max()) <= 2e-5 * scale
    assert float((Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max()) <= 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 64, 65, 200, 1030])
def test_block_reflector_back_transformation(hip, n):
    """eigh_native.ormtr_native applies Q = H_0 ... H_{n-2} of clo_sytrd_f32: Q T Q^T reproduces the input and
    Q is orthogonal (applied to the identity)."""
    from curvlinops_amd import eigh_native

    dev = torch.device("cuda:0")
    A64 = _sym_case("indefinite", n, dev)
    ld = (n + 3) // 4 * 4
    work = torch.zeros(n, ld, device=dev)
    work[:, :n] = A64.float().to(dev)
    D, E, tau = hip.sytrd_(work, n)
    Zr = torch.zeros(n, ld, device=dev)
    Zr[:, :n] = torch.eye(n, device=dev)
    eigh_native.ormtr_native(work, tau, Zr, n)          # rows of Zr = columns of Q ...
    Q = Zr[:, :n].T.double().cpu()
    T = torch.diag(D.double().cpu()) + torch.diag(E[: n - 1].double().cpu(), 1) + torch.diag(E[: n - 1].double().cpu(), -1)
    sc = float(A64.abs().max())
    assert float((Q.T @ Q - torch.eye(n, dtype=torch.float64)).abs().max()) <= 1e-5
    assert float((Q @ T @ Q.T - A64).abs().max()) <= 2e-5 * sc * max(1.0, n ** 0.5 / 8)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 577, 1153, 2305, 4609])
def test_native_eigh_at_the_factor_sizes_of_the_benchmarks(hip, n):
    """The hand-written solver end to end (no torch.linalg.eigh / rocSOLVER on the way) at the factor orders of
    ResNet-18 / the encoder: |Q^T Q - I| <= 1e-5 and Q diag(lam) Q^T against the float64 matrix within 4 eps32 |A|_2
    (the backward-error scale of an fp32 eigensolver) and, up to n = 577, within 1e-4 |A|max entrywise.  Beyond, the
    entrywise error of this rank-one-dominated matrix (n = 4609: |A|_2 = 3500 |A|max, ONE rounding of the top
    eigenvalue is 2e-4 |A|max) depends on the order of the reduction's sums: round 3's column launches 0.7e-4 at 4609,
    the persistent panel launches 0.7 ... 2.3e-4 for 32 ... 128 workgroups (tools/cmp_sytrd_recon.py), and since a group's
    partial sums are added in arrival order (float atomics) also on the run: 0.5 ... 1.4e-4 |A|max at n = 1153 -- all below
    half an eps32 |A|_2."""
    from curvlinops_amd import linalg_native as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    X = torch.rand(max(16, n // 3), n, generator=g, dtype=torch.float64)      # rank-deficient covariance
    A64 = X.T @ X / X.shape[0]
    A = A64.to(dev, torch.float32)
    lam, Q = L.eigh(A)
    Qd, ld_ = Q.double().cpu(), lam.double().cpu()
    assert float((Qd.T @ Qd - torch.eye(n, dtype=torch.float64)).abs().max()) <= 1e-5
    rec = float(((Qd * ld_) @ Qd.T - A.double().cpu()).abs().max())
    assert rec <= 4.0 * 2.0 ** -24 * float(ld_.abs().max())
    if n <= 577:
        assert rec <= 1e-4 * float(A64.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [577, 1153, 2305, 4609])
def test_native_eigh_accepts_full_rank_factors_without_float64_retry(hip, n):
    """Well-conditioned ("Wishart", r = 2 n rows) covariances -- what input covariances become once the batch has more
    rows than features -- have ||A||_2 ~ 0.3 n max|A|: the acceptance test of the float32 result is relative to the
    matrix norm (``linalg_native._residual_tol``), so a CORRECT result is accepted and no float64 vendor solve runs
    (round 4: absolute bound, every such factor beyond n = 577 was decomposed twice).  Accuracy against float64 LAPACK:
    orthogonality 1e-5, reconstruction within 16 eps32 ||A||_2 (measured 5 ... 10), spectrum within 2 sqrt(n) eps32 ||A||_2
    (the backward-error scale of a one-stage float32 reduction; measured 17 eps32 at n = 577)."""
    from curvlinops_amd import linalg_native as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    X = torch.rand(2 * n, n, generator=g, dtype=torch.float64)
    A64 = X.T @ X / X.shape[0]
    A = A64.to(dev, torch.float32)
    before = L.FLOAT64_RETRIES
    lam, Q = L.eigh(A)
    torch.cuda.synchronize()
    assert L.FLOAT64_RETRIES == before, "a correct float32 result was rejected and redone in float64"
    Qd, ld_ = Q.double().cpu(), lam.double().cpu()
    ref = torch.linalg.eigvalsh(A.double().cpu())
    top = float(ref.abs().max())
    assert float((Qd.T @ Qd - torch.eye(n, dtype=torch.float64)).abs().max()) <= 1e-5
    rec = float(((Qd * ld_) @ Qd.T - A.double().cpu()).abs().max())
    assert rec <= 16.0 * 2.0 ** -23 * top, f"reconstruction {rec / (2.0 ** -23 * top):.1f} eps32 |A|_2"
    spec = float((ld_ - ref).abs().max())
    assert spec <= 2.0 * n ** 0.5 * 2.0 ** -23 * top, f"spectrum {spec / (2.0 ** -23 * top):.1f} eps32 |A|_2"


@pytest.mark.gpu
def test_native_eigh_of_a_non_finite_factor_does_not_fault(hip):
    """A NaN in the factor (diverged training): the device-side rank sort of the divide & conquer must still emit a
    permutation (NaN keys sort last), so the later gathers never index with uninitialised workspace; the result is
    non-finite or the float64 retry's, never an out-of-bounds access, and the next healthy call is unaffected."""
    from curvlinops_amd import linalg_native as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    X = torch.rand(200, 320, generator=g)
    A = (X.T @ X / 200).to(dev)
    bad = A.clone()
    bad[7, 9] = bad[9, 7] = float("nan")
    try:
        lam, Q = L.eigh(bad)
        torch.cuda.synchronize()
        assert not bool(torch.isfinite(lam).all()) or not bool(torch.isfinite(Q).all())
    except (RuntimeError, torch.linalg.LinAlgError):   # the float64 retry may refuse non-finite input: fine
        torch.cuda.synchronize()
    lam, Q = L.eigh(A)
    assert float((A @ Q - Q * lam).abs().max()) <= 1e-4 * float(lam.abs().max())


def _assert_same_reduction(got, ref, n, tol=2e-6):
    """Two runs of clo_sytrd_f32 on the same matrix.  A group's partial sums are added with float atomics (arrival
    order), and the map A -> T is ill-conditioned for rank-deficient A (entries of T past the numerical rank move by
    1e-2 under a rounding-level change), so the runs are compared through what IS well conditioned: the spectrum of T."""
    from scipy.linalg import eigvalsh_tridiagonal

    def spectrum(red):
        d, e = red[0][:n].double().cpu().numpy(), red[1][: n - 1].double().cpu().numpy()
        return eigvalsh_tridiagonal(d, e)

    lam, lam_ref = spectrum(got), spectrum(ref)
    assert np.isfinite(lam).all()
    assert np.abs(lam - lam_ref).max() <= tol * np.abs(lam_ref).max() * n ** 0.5


@pytest.mark.gpu
def test_persistent_grids_of_different_streams_are_admitted_safely(hip):
    """Round 4: panel launches of clo_sytrd_f32 (up to 256 workgroups that wait for each other) on two streams next to
    persistent GGN products (256 workgroups) on a third.  Partly resident persistent grids would wait for each other
    forever; the library's admission control (csrc/persist_gate.h) makes a launch wait, device side, until everything
    in flight plus itself fits the chip.  Products equal the serial ones bit for bit, reductions to rounding; nothing hangs or traps."""
    device = torch.device("cuda:0")
    n = 1100
    ld = (n + 3) // 4 * 4
    mats = [_sym_case(kind, n, device).float().to(device) for kind in ("lowrank", "indefinite")]

    def reduce(A):
        P = torch.zeros(n, ld, device=device)
        P[:, :n] = A
        return hip.sytrd_(P, n)

    serial = [reduce(A) for A in mats]
    g = np.random.default_rng(5)
    dims, acts, N = [1024, 2688, 2688, 10], ["relu", "relu", "identity"], 8
    Ws, bs, vWs, vbs, X, y = _mega_case(g, dims, acts, N, "mse")
    plan = hip.MLPPlan(dims, [ACT_CODE[a] for a in acts])
    dW, db, dVW, dVb, dX = [dev(W) for W in Ws], [dev(b) for b in bs], [dev(v) for v in vWs], [dev(v) for v in vbs], dev(X)
    scale = 2.0 * O.reduction_factor("mse", "mean", N, dims[-1])

    def product():
        oW, ob = [torch.full_like(w, float("nan")) for w in dW], [torch.full_like(b, float("nan")) for b in db]
        plan.ggn_matvec(dW, db, dVW, dVb, oW, ob, dX, 0, scale, 1.0, 0.0)
        return oW + ob

    ref = product()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    with torch.cuda.stream(streams[2]):
        plan.workspace(N, dX.device)
    torch.cuda.synchronize()
    got_red, got_prod = [[], []], []
    for rep in range(3):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                got_red[i].append(reduce(mats[i]))
        with torch.cuda.stream(streams[2]):
            for _ in range(10):
                got_prod.append(product())
    torch.cuda.synchronize()
    # (a torn or stale exchange moves the spectrum at the 1e-2 level; arrival-order rounding does not)
    for i in range(2):
        for D, E, tau in got_red[i]:
            _assert_same_reduction((D, E, tau), serial[i], n)
    for out in got_prod:
        assert all(torch.equal(a, b) for a, b in zip(out, ref))


@pytest.mark.gpu
def test_sytrd_is_deterministic(hip):
    """Rank-deficient factors take the per-block panel pass in many columns: no block may observe another
    block's updates of the panel (a race here showed up as run-to-run differences of 1e-2).  Runs agree to rounding
    (round 4: a group's partial sums are added with float atomics, in arrival order)."""
    dev = torch.device("cuda:0")
    n = 777
    A64 = _sym_case("lowrank", n, dev)
    ld = (n + 3) // 4 * 4
    outs = []
    for _ in range(4):
        P = torch.zeros(n, ld, device=dev)
        P[:, :n] = A64.float().to(dev)
        D, E, tau = hip.sytrd_(P, n)
        outs.append((D.clone(), E.clone(), tau.clone()))
    for red in outs[1:]:
        _assert_same_reduction(red, outs[0], n)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["eigh", "eigh_many", "eigh_sytrd"])
def test_eigh_orthogonal_on_dead_relu_factor(hip, entry):
    """A real KFAC input covariance (845 flattened ReLU features of a small conv net, 338 dead units, repeated
    rows; produced by tools/fuzz_kfac.py seed 3 case 59) on which rocSOLVER's ssyevd -- plain torch.linalg.eigh --
    returns eigenvectors with |Q^T Q - I| = 0.07 (0.27 on the normalised matrix) on this platform.  Every
    entry point of the package must return an orthogonal basis (verified + decomposed again when needed)."""
    from curvlinops_amd import linalg_native as L

    dev = torch.device("cuda:0")
    A = torch.as_tensor(load_golden("eigh_regression")["dead_relu_845"]["factor"]).to(dev)
    n = A.shape[0]
    if entry == "eigh":
        lam, Q = L.eigh(A)
    elif entry == "eigh_sytrd":
        lam, Q = L.eigh_sytrd(A)
    else:
        (lam, Q), (lam2, Q2) = L.eigh_many([A, 2.0 * A])   # equal sizes: the stacked batched call
        assert float((Q2.T @ Q2 - torch.eye(n, device=dev)).abs().max()) <= 2e-5
    A64, Q64, l64 = A.double().cpu(), Q.double().cpu(), lam.double().cpu()
    assert float((Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max()) <= 2e-5
    ref = torch.linalg.eigvalsh(A64)
    # residual on the backward-error scale of fp32: eps32 |A|_2 = 6e-8 x 256 x |A|max = 1.5e-5 |A|max here; the entrywise residual moves
    # with the summation order of the reduction (round 3: 2.2e-5 ... 3.0e-5 |A|max, persistent panels 3.2e-5 at most)
    assert float((A64 @ Q64 - Q64 * l64).abs().max()) <= 8.0 * 2.0 ** -24 * float(ref.abs().max())
    assert float((l64 - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.gpu
def test_fuzz_eigh(hip):
    """Random orders / spectra / scales through both eigensolver routes (tools/fuzz_eigh.py)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_eigh

    worst, failures = fuzz_eigh.run(seed=5, ncase=60)
    assert not failures, "\n".join(failures)
    os.environ["CLO_FUZZ_EIGH_DEFAULT"] = "1"
    try:
        worst, failures = fuzz_eigh.run(seed=6, ncase=60)
    finally:
        del os.environ["CLO_FUZZ_EIGH_DEFAULT"]
    assert not failures, "\n".join(failures)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dead", [(64, 0.5), (300, 0.9), (845, 0.3), (97, 1.0), (33, 0.0)])
@pytest.mark.parametrize("entry", ["eigh", "eigh_many"])
def test_eigh_deflates_zero_rows(hip, n, dead, entry):
    """Exactly-zero rows / columns (dead ReLU features) are split off before the solver runs: the result is a
    complete orthonormal eigenbasis of the full matrix with ascending eigenvalues (indefinite input, so the zero
    eigenvalues sit in the middle of the spectrum)."""
    from curvlinops_amd import linalg_native as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    A64 = _sym_case("indefinite", n, dev)
    kill = torch.rand(n, generator=g) < dead
    A64[kill, :] = 0.0
    A64[:, kill] = 0.0
    A = A64.to(dev, torch.float32)
    if entry == "eigh":
        lam, Q = L.eigh(A)
    else:
        (lam, Q), (lam2, Q2) = L.eigh_many([A, _sym_case("full", n, dev).to(dev, torch.float32)])
        assert float((Q2.T @ Q2 - torch.eye(n, device=dev)).abs().max()) <= 2e-5
    A32, l64, Q64 = A.double().cpu(), lam.double().cpu(), Q.double().cpu()
    ref = torch.linalg.eigvalsh(A32)
    scale = max(float(A32.abs().max()), 1.0)
    assert torch.all(l64[1:] >= l64[:-1])
    assert float((l64 - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1e-30) + 1e-12
    assert float((A32 @ Q64 - Q64 * l64).abs().max()) <= 2e-5 * scale
    assert float((Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max()) <= 2e-5
    assert int((l64 == 0).sum()) >= int(kill.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 5, 32])
@pytest.mark.parametrize("shape", [(64, 65, 10, 11), (7, 7, 3, 3), (130, 128, 577, 577), (1, 4, 9, 1), (512, 512, 129, 129)])
def test_kron_single_call_entries(hip, shape, K):
    """clo_kron_matmat / clo_eigh_apply / clo_kron_matmat_blocks (csrc/kron.hip; reference kronecker.py:141-171,
    eigh.py:84-105) on K-major operands against the float64 einsum the reference evaluates."""
    A, a, B, b = shape
    g = torch.Generator().manual_seed(A * 7 + a + K)
    S1, S2 = torch.rand(A, a, generator=g, dtype=torch.float64) - 0.5, torch.rand(B, b, generator=g, dtype=torch.float64) - 0.5
    X = torch.rand(K, a, b, generator=g, dtype=torch.float64) - 0.5
    ref = torch.einsum("Aa,kab,Bb->kAB", S1, X, S2).reshape(K, A * B)
    d = lambda t: t.float().cuda().contiguous()   # noqa: E731
    y = hip.kron_matmat(d(S1), d(S2), d(X).reshape(K, a * b), K)
    assert rel_err(y.cpu(), ref) < TOL
    Xt = torch.rand(K, A, B, generator=g, dtype=torch.float64) - 0.5
    ref_t = torch.einsum("Aa,kAB,Bb->kab", S1, Xt, S2).reshape(K, a * b)
    yt = hip.kron_matmat(d(S1), d(S2), d(Xt).reshape(K, A * B), K, trans=3)
    assert rel_err(yt.cpu(), ref_t) < TOL
    # eigen-decomposed block with a Kronecker eigenbasis
    Q1, Q2 = torch.linalg.qr(torch.rand(A, A, generator=g, dtype=torch.float64))[0], torch.linalg.qr(torch.rand(B, B, generator=g, dtype=torch.float64))[0]
    lam = torch.rand(A * B, generator=g, dtype=torch.float64) + 0.1
    Xe = torch.rand(K, A, B, generator=g, dtype=torch.float64) - 0.5
    Z = torch.einsum("Aa,kAB,Bb->kab", Q1, Xe, Q2) * lam.reshape(1, A, B)
    ref_e = torch.einsum("Aa,kab,Bb->kAB", Q1, Z, Q2).reshape(K, A * B)
    ye = hip.eigh_apply(d(Q1), d(Q2), d(lam), d(Xe).reshape(K, A * B), K)
    assert rel_err(ye.cpu(), ref_e) < TOL
    for rows in (1, 2, 3):   # eigenvectors of factor 1 / 2 / both in the rows of the arrays passed
        yr = hip.eigh_apply(d(Q1.T if rows & 1 else Q1), d(Q2.T if rows & 2 else Q2), d(lam), d(Xe).reshape(K, A * B), K, rows=rows)
        assert rel_err(yr.cpu(), ref_e) < TOL
    # mixed orientation, padded leading dimension
    S1p = torch.zeros(A, a + 3, dtype=torch.float64); S1p[:, :a] = S1
    ym = hip.kron_matmat(d(S1p)[:, :a], d(S2.T), d(X).reshape(K, a * b), K, trans=2)
    assert rel_err(ym.cpu(), ref) < TOL
    # the three of them as blocks of one call (a plain block twice, then eigen-decomposed ones in their own call)
    ys = hip.kron_blocks([(d(S1), d(S2), None, 0), (d(S2), d(S1), None, 0)], [d(X).reshape(K, a * b), d(X.transpose(1, 2)).reshape(K, a * b)], K)
    assert rel_err(ys[0].cpu(), ref) < TOL
    assert rel_err(ys[1].cpu(), torch.einsum("Bb,kba,Aa->kBA", S2, X.transpose(1, 2), S1).reshape(K, A * B)) < TOL
    (yb,) = hip.kron_blocks([(d(Q1), d(Q2), d(lam), 0)], [d(Xe).reshape(K, A * B)], K)
    assert torch.equal(yb, ye)


@pytest.mark.gpu
@pytest.mark.parametrize("V,B,S,d1,d2", [(1, 16, 1, 10, 33), (2, 8, 5, 12, 27), (3, 4, 49, 64, 148), (1, 64, 1, 512, 129)])
@pytest.mark.parametrize("rows", [0, 1, 2, 3])
def test_ekfac_correction_single_call(hip, V, B, S, d1, d2, rows):
    """clo_ekfac_correction_f32 (reference computers/ekfac_hooks.py:25-238): sum_{v,n} (Qg^T (sum_s g a^T) Qa)^2 against the
    float64 einsum over materialised per-example gradients; eigenvector arrays in either orientation."""
    gen = torch.Generator().manual_seed(V * 100 + S)
    g = torch.rand(V, B, S, d1, generator=gen, dtype=torch.float64) - 0.5
    a = torch.rand(B, S, d2, generator=gen, dtype=torch.float64) - 0.5
    Qg = torch.linalg.qr(torch.rand(d1, d1, generator=gen, dtype=torch.float64))[0]
    Qa = torch.linalg.qr(torch.rand(d2, d2, generator=gen, dtype=torch.float64))[0]
    per = torch.einsum("vnsi,nsj->vnij", g @ Qg, a @ Qa)
    ref = per.square().sum(dim=(0, 1))
    d = lambda t: t.float().cuda().contiguous()   # noqa: E731
    out = hip.ekfac_correction(d(g), d(Qg.T if rows & 1 else Qg), d(a), d(Qa.T if rows & 2 else Qa), rows=rows)
    assert rel_err(out.cpu(), ref) < 5 * TOL
