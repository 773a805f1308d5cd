"""Ownership rules of the drop-in boundary (reference ``test/test_kfac.py:1026-1085``,
``test/test_ekfac.py``): building an operator must not touch ``.grad`` nor leave the module's
parameters replaced, and (E)KFAC / KFOC operators must survive ``torch.save`` / ``torch.load``."""

import pytest
import torch
from torch import nn

import curvlinops_amd as C

CLASSES = [C.KFACLinearOperator, C.EKFACLinearOperator, C.KFOCLinearOperator]


def _check(cls, device, tmp_path):
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(3, 4), nn.Tanh(), nn.Linear(4, 2)).to(device)
    params = dict(model.named_parameters())
    X, y = torch.rand(5, 3, device=device), torch.rand(5, 2, device=device)
    for p in params.values():
        p.grad = torch.rand_like(p)
    grads_before = [p.grad.clone() for p in params.values()]
    values_before = [p.detach().clone() for p in params.values()]
    kw = {"fisher_type": "type-2"} if cls is C.KFOCLinearOperator else {}
    op = cls(model, nn.MSELoss(), params, [(X, y)], **kw)
    for g0, v0, (name, p) in zip(grads_before, values_before, params.items()):
        assert torch.equal(g0, p.grad), name
        assert torch.equal(v0, p.detach()), name
        assert dict(model.named_parameters())[name] is p  # the module still owns the same tensors
    eye = torch.eye(op.shape[1], device=device)
    before = op @ eye
    path = tmp_path / "linop.pt"
    torch.save(op, path)
    loaded = torch.load(path, weights_only=False)
    assert torch.equal(before, loaded @ eye)
    inv = loaded.inverse(damping=1e-2) if cls is not C.KFOCLinearOperator else None
    if inv is not None:
        assert torch.allclose(inv @ eye, op.inverse(damping=1e-2) @ eye, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cls", CLASSES, ids=lambda c: c.__name__)
def test_build_leaves_grads_alone_and_operator_pickles_cpu(cls, tmp_path):
    _check(cls, torch.device("cpu"), tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("cls", CLASSES, ids=lambda c: c.__name__)
def test_build_leaves_grads_alone_and_operator_pickles_gpu(cls, tmp_path):
    _check(cls, torch.device("cuda:0"), tmp_path)
