"""Pin the CPU oracle (oracle/mlp_numpy.py) against vectors produced by the reference itself."""

import numpy as np
import pytest

from conftest import load_golden, mlp_case_tensors
from oracle import mlp_numpy as O

CASES = sorted(load_golden("mlp_curvature"))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", ["ggn", "hessian", "ef"])
def test_mlp_oracle_matches_reference(golden_mlp, case, kind):
    rec = golden_mlp[case]
    dims, acts, bias, loss, red, Ws, bs, data = mlp_case_tensors(rec)
    shapes = [W.shape for W in Ws]
    for vec, ref in ((rec["v"], rec[f"{kind}_v"]), (rec["V"], rec[f"{kind}_V"])):
        cols = vec.reshape(vec.shape[0], -1)
        refc = ref.reshape(ref.shape[0], -1)
        for k in range(cols.shape[1]):
            vWs, vbs = O.unflatten_params(cols[:, k], shapes, bias)
            oW, ob = O.matvec(kind, Ws, bs, acts, data, loss, red, vWs, vbs)
            got = O.flatten_params(oW, ob)
            scale = np.abs(refc[:, k]).max()
            assert np.abs(got - refc[:, k]).max() <= 1e-10 * max(scale, 1e-30) + 1e-14


@pytest.mark.parametrize("case", sorted(load_golden("mlp_columns")))
@pytest.mark.parametrize("kind", ["ggn", "hessian", "ef"])
def test_mlp_oracle_matches_reference_columns(case, kind):
    """K = 8 columns on the shapes the native column kernels accept (tests/golden/mlp_columns.npz, generated from the
    reference by oracle/make_golden.py columns)."""
    rec = load_golden("mlp_columns")[case]
    dims, acts, bias, loss, red, Ws, bs, data = mlp_case_tensors(rec)
    shapes = [W.shape for W in Ws]
    V, ref = rec["V"], rec[f"{kind}_V"]
    for k in range(V.shape[1]):
        vWs, vbs = O.unflatten_params(V[:, k], shapes, bias)
        oW, ob = O.matvec(kind, Ws, bs, acts, data, loss, red, vWs, vbs)
        got = O.flatten_params(oW, ob)
        assert np.abs(got - ref[:, k]).max() <= 1e-10 * max(np.abs(ref[:, k]).max(), 1e-30) + 1e-14
