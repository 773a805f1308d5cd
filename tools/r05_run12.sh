R=$PWD; OUT=$R/gpurun_out/r05_run12; mkdir -p $OUT
python -m pytest tests -x -q -m gpu -k "hessian or Hessian or updates or merged or golden or abi" > $OUT/hess_tests.txt 2>&1; tail -4 $OUT/hess_tests.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/hess_timing.txt
import time, torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
for dims, N in (([128, 256, 64, 10], 64), ([1024, 2688, 2688, 10], 8), ([1024, 2688, 2688, 10], 128)):
    m = nn.Sequential(nn.Linear(dims[0], dims[1]), nn.Tanh(), nn.Linear(dims[1], dims[2]), nn.Tanh(), nn.Linear(dims[2], dims[3])).to(dev)
    X, y = torch.rand(N, dims[0], device=dev), torch.randint(0, dims[3], (N,), device=dev)
    op = C.HessianLinearOperator(m, nn.CrossEntropyLoss(), dict(m.named_parameters()), [(X, y)], check_deterministic=False)
    v = torch.rand(op.shape[1], device=dev)
    for frozen in (False, True):
        op.assume_frozen = frozen
        for _ in range(10): op @ v
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): op @ v
        torch.cuda.synchronize(); print(f"dims {dims} N={N} Hessian (assume_frozen={frozen}): {1e4*(time.perf_counter()-t0):.1f} us per product", flush=True)
PY
