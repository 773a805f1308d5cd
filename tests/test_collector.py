"""The "collector" KFAC backend (alias "make_fx": the reference's second backend, `kfac.py:89-92`): factors
from the affine operations the forward pass executes -- functional models and weight tying.  The weight-tying
tests restate the reference's own (`test/test_kfac.py:199-360`): with one datum KFAC-expand (type-2) equals the
exact block-diagonal GGN when every use of a tied weight is counted as an extra weight-sharing position."""

import pytest
import torch
import torch.nn.functional as F
from torch import nn

import curvlinops_amd as C
from conftest import load_golden
from helpers import KFAC_MODELS, LOSS, golden_data, load_into, rel_err

F64 = torch.float64


class SplitConcat(nn.Module):
    """The same Linear applied to the two halves of the input (`test/utils.py:346-376`)."""

    def __init__(self, D, bias):
        super().__init__()
        self.linear = nn.Linear(D, D, bias=bias)
        self.D = D

    def forward(self, x):
        x1, x2 = x.split(self.D, dim=-1)
        return torch.cat([self.linear(x1), self.linear(x2)], dim=-1)


class TiedSplitConcat(nn.Module):
    """Two Linear modules with a tied weight and independent biases (`test/utils.py:380-414`)."""

    def __init__(self, D, bias1, bias2):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(D, D, bias=bias1), nn.Linear(D, D, bias=bias2)
        self.linear2.weight = self.linear1.weight
        self.D = D

    def forward(self, x):
        x1, x2 = x.split(self.D, dim=-1)
        return torch.cat([self.linear1(x1), self.linear2(x2)], dim=-1)


def block_diagonal_ggn(model, loss, params, data, mapping):
    """Dense GGN restricted to the blocks of the parameter groups in `mapping`."""
    G = C.GGNLinearOperator(model, loss, params, data, check_deterministic=False)
    dense = G @ torch.eye(G.shape[1], dtype=F64)
    starts, pos = {}, 0
    for n, p in params.items():
        starts[n] = (pos, pos + p.numel())
        pos += p.numel()
    mask = torch.zeros_like(dense, dtype=torch.bool)
    for group in mapping:
        idx = torch.cat([torch.arange(*starts[n]) for n in group.values()])
        mask[idx[:, None], idx[None, :]] = True
    return dense * mask


@pytest.mark.parametrize("backend", ["collector", "make_fx"])
@pytest.mark.parametrize("reduction", ["mean", "sum"])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("sep", [True, False])
def test_weight_tying_type2_equals_exact_ggn(backend, reduction, bias, sep):
    torch.manual_seed(0)
    D = 4
    model = SplitConcat(D, bias).to(F64)
    params = dict(model.named_parameters())
    data = [(torch.rand(1, 2 * D, dtype=F64), torch.rand(1, 2 * D, dtype=F64))]
    loss = nn.MSELoss(reduction=reduction)
    K = C.KFACLinearOperator(model, loss, params, data, fisher_type="type-2", kfac_approx="expand",
                             separate_weight_and_bias=sep, backend=backend)
    mapping = ([{"W": "linear.weight"}, {"b": "linear.bias"}] if sep and bias else
               [{"W": "linear.weight", "b": "linear.bias"}] if bias else [{"W": "linear.weight"}])
    ref = block_diagonal_ggn(model, loss, params, data, mapping)
    assert rel_err(K @ torch.eye(K.shape[1], dtype=F64), ref.numpy()) < 1e-10
    E = C.EKFACLinearOperator(model, loss, params, data, fisher_type="type-2", separate_weight_and_bias=sep,
                              backend=backend)  # `test/test_ekfac.py:195-204`
    assert rel_err(E @ torch.eye(K.shape[1], dtype=F64), ref.numpy()) < 1e-8
    hooks = C.KFACLinearOperator(model, loss, params, data, fisher_type="type-2", separate_weight_and_bias=sep,
                                 backend="hooks")
    assert rel_err(hooks @ torch.eye(K.shape[1], dtype=F64), ref.numpy()) > 1e-3  # module hooks: wrong scaling


@pytest.mark.parametrize("reduction", ["mean", "sum"])
@pytest.mark.parametrize("sep", [True, False])
def test_mixed_bias_weight_tying(reduction, sep):
    torch.manual_seed(0)
    D = 4
    model = TiedSplitConcat(D, True, False).to(F64)
    params = dict(model.named_parameters())
    data = [(torch.rand(1, 2 * D, dtype=F64), torch.rand(1, 2 * D, dtype=F64))]
    loss = nn.MSELoss(reduction=reduction)
    K = C.KFACLinearOperator(model, loss, params, data, fisher_type="type-2", separate_weight_and_bias=sep,
                             backend="collector")
    mapping = [{"W": "linear1.weight"}, {"b": "linear1.bias"}] if sep else [{"W": "linear1.weight", "b": "linear1.bias"}]
    ref = block_diagonal_ggn(model, loss, params, data, mapping)
    assert rel_err(K @ torch.eye(K.shape[1], dtype=F64), ref.numpy()) < 1e-10


def test_conflicting_biases_raise_under_joint_treatment():
    torch.manual_seed(0)
    model = TiedSplitConcat(4, True, True)
    data = [(torch.rand(1, 8), torch.rand(1, 8))]
    with pytest.raises(ValueError, match="conflicting biases"):
        C.KFACLinearOperator(model, nn.MSELoss(), dict(model.named_parameters()), data, fisher_type="type-2",
                             separate_weight_and_bias=False, backend="collector")


def test_functional_model_with_reused_weight():
    """A callable `(params, X) -> prediction`: two layers share W (and its bias), plus a head."""
    torch.manual_seed(0)
    params = {"W": torch.randn(5, 5, dtype=F64) * 0.4, "b": torch.randn(5, dtype=F64) * 0.1,
              "head": torch.randn(3, 5, dtype=F64) * 0.4}

    def f(p, X):
        h = torch.tanh(F.linear(X, p["W"], p["b"]))
        h = torch.tanh(F.linear(h, p["W"], p["b"]))
        return F.linear(h, p["head"])

    data = [(torch.rand(6, 5, dtype=F64), torch.rand(6, 3, dtype=F64)), (torch.rand(4, 5, dtype=F64), torch.rand(4, 3, dtype=F64))]
    K = C.KFACLinearOperator(f, nn.MSELoss(), params, data, fisher_type="type-2", separate_weight_and_bias=False,
                             backend="collector")
    (Gw, Aw), (Gh, Ah) = [tuple(blk) for blk in K[1]]
    # by hand: both uses of W are weight-sharing positions
    with torch.no_grad():
        a_all, N = [], 10
        for X, _ in data:
            h1 = torch.tanh(F.linear(X, params["W"], params["b"]))
            a_all.append(torch.stack([X, h1], dim=1))  # [B, 2 uses, 5]
        a = torch.cat(a_all)
        a1 = torch.cat([a, torch.ones(*a.shape[:2], 1, dtype=F64)], dim=-1).reshape(-1, 6)
        assert rel_err(Aw, (a1.T @ a1 / (N * 2)).numpy()) < 1e-12
    assert Gw.shape == (5, 5) and Ah.shape == (5, 5) and Gh.shape == (3, 3)
    assert not any(p.grad is not None for p in params.values())


@pytest.mark.parametrize("case", sorted(KFAC_MODELS))
def test_collector_equals_hooks_backend_on_module_models(case):
    rec = load_golden("kfac")[case]
    loss, red = str(rec["loss"]), str(rec["reduction"])
    model = KFAC_MODELS[case]()
    params = load_into(model, rec, F64, "cpu")
    data = golden_data(rec, F64, "cpu", loss)
    approxes = ["expand", "reduce"] if case.startswith(("cnn", "seq")) else ["expand"]
    for fisher in ("type-2", "empirical", "forward-only"):
        for approx in approxes:
            for sep in (True, False):
                tag = f"{fisher}|{approx}|{'sep' if sep else 'joint'}"
                K = C.KFACLinearOperator(model, LOSS[loss](reduction=red), params, data, fisher_type=fisher,
                                         kfac_approx=approx, separate_weight_and_bias=sep, backend="collector")
                for b, block in enumerate(K[1]):
                    for f, fac in enumerate(block):
                        assert rel_err(fac, rec[f"{tag}/block{b}_factor{f}"]) < 1e-10, (case, tag, b, f)
                assert rel_err(K @ torch.as_tensor(rec["V"]), rec[f"{tag}/KV"]) < 1e-10
    assert all(p.grad is None for p in model.parameters())


@pytest.mark.parametrize("case", [c for c in sorted(KFAC_MODELS) if not c.startswith("seq")])
def test_collector_ekfac_equals_reference_goldens(case):
    rec = load_golden("kfac")[case]
    loss, red = str(rec["loss"]), str(rec["reduction"])
    model = KFAC_MODELS[case]()
    params = load_into(model, rec, F64, "cpu")
    data = golden_data(rec, F64, "cpu", loss)
    V = torch.as_tensor(rec["V"])
    for tag in sorted({k.split("/")[0] for k in rec if k.startswith("ekfac")}):
        _, fisher, sep = tag.split("|")
        E = C.EKFACLinearOperator(model, LOSS[loss](reduction=red), params, data, fisher_type=fisher,
                                  separate_weight_and_bias=sep == "sep", backend="collector")
        assert rel_err(E @ V, rec[f"{tag}/EV"]) < 1e-8, (case, tag)
        assert rel_err(E.trace(), rec[f"{tag}/trace"]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mlp_mse_mean", "cnn_ce_mean"])
def test_collector_backend_gpu(case):
    """fp32 on the MI355X: the tapped inputs / output-gradients go through the same HIP SYRK kernels."""
    from curvlinops_amd import _hip

    _hip.load()
    dev = torch.device("cuda:0")
    rec = load_golden("kfac")[case]
    loss, red = str(rec["loss"]), str(rec["reduction"])
    model = KFAC_MODELS[case]()
    params = load_into(model, rec, torch.float32, dev)
    data = golden_data(rec, torch.float32, dev, loss)
    V = torch.as_tensor(rec["V"], dtype=torch.float32).to(dev)
    for fisher in ("type-2", "empirical"):
        for sep in (True, False):
            tag = f"{fisher}|expand|{'sep' if sep else 'joint'}"
            K = C.KFACLinearOperator(model, LOSS[loss](reduction=red), params, data, fisher_type=fisher,
                                     separate_weight_and_bias=sep, backend="collector")
            for b, block in enumerate(K[1]):
                for f, fac in enumerate(block):
                    assert rel_err(fac, rec[f"{tag}/block{b}_factor{f}"]) < 1e-4, (case, tag, b, f)
            assert rel_err(K @ V, rec[f"{tag}/KV"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("sep", [True, False])
def test_weight_tying_fp32_gpu(bias, sep):
    """The reference's weight-tying test (`test/test_kfac.py:273-360`) in float32 on the device: the collector
    backend's factors come from the HIP SYRK kernels; with one datum type-2 KFAC / EKFAC equal the exact
    block-diagonal GGN (float64 CPU) to fp32 accuracy."""
    from curvlinops_amd import _hip

    _hip.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    D = 4
    model64 = SplitConcat(D, bias).to(F64)
    params64 = dict(model64.named_parameters())
    X, y = torch.rand(1, 2 * D, dtype=F64), torch.rand(1, 2 * D, dtype=F64)
    mapping = ([{"W": "linear.weight"}, {"b": "linear.bias"}] if sep and bias else
               [{"W": "linear.weight", "b": "linear.bias"}] if bias else [{"W": "linear.weight"}])
    for reduction in ("mean", "sum"):
        loss = nn.MSELoss(reduction=reduction)
        ref = block_diagonal_ggn(model64, loss, params64, [(X, y)], mapping).numpy()
        model = SplitConcat(D, bias).to(dev)
        model.load_state_dict({k: v.float() for k, v in model64.state_dict().items()})
        params = dict(model.named_parameters())
        data = [(X.float().to(dev), y.float().to(dev))]
        eye = torch.eye(ref.shape[0], device=dev)
        K = C.KFACLinearOperator(model, loss, params, data, fisher_type="type-2", kfac_approx="expand",
                                 separate_weight_and_bias=sep, backend="collector")
        assert rel_err(K @ eye, ref) < 1e-4, reduction
        E = C.EKFACLinearOperator(model, loss, params, data, fisher_type="type-2", separate_weight_and_bias=sep,
                                  backend="collector")
        assert rel_err(E @ eye, ref) < 1e-3, reduction
        # two tied modules with different bias settings (`test/utils.py:380-414`), joint and separate
    model64 = TiedSplitConcat(D, True, False).to(F64)
    params64 = dict(model64.named_parameters())
    mapping = [{"W": "linear1.weight"}, {"b": "linear1.bias"}] if sep else [{"W": "linear1.weight", "b": "linear1.bias"}]
    loss = nn.MSELoss()
    ref = block_diagonal_ggn(model64, loss, params64, [(X, y)], mapping).numpy()
    model = TiedSplitConcat(D, True, False).to(dev)
    model.load_state_dict({k: v.float() for k, v in model64.state_dict().items()})
    model.linear2.weight = model.linear1.weight
    K = C.KFACLinearOperator(model, loss, dict(model.named_parameters()), [(X.float().to(dev), y.float().to(dev))],
                             fisher_type="type-2", separate_weight_and_bias=sep, backend="collector")
    assert rel_err(K @ torch.eye(ref.shape[0], device=dev), ref) < 1e-4


def test_view_of_tracked_parameter_is_refused():
    """A transposed / viewed tied weight must not be dropped silently (it would be missing from the factors)."""

    class TransposedTie(nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = nn.Linear(4, 4, bias=False)

        def forward(self, x):
            return F.linear(self.enc(x), self.enc.weight.T)

    model = TransposedTie().to(F64)
    data = [(torch.rand(2, 4, dtype=F64), torch.rand(2, 4, dtype=F64))]
    with pytest.raises(NotImplementedError, match="view of tracked parameter"):
        C.KFACLinearOperator(model, nn.MSELoss(), dict(model.named_parameters()), data, fisher_type="type-2",
                             backend="collector")
