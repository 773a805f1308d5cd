R=$PWD; OUT=$R/gpurun_out/r05_run11; mkdir -p $OUT
python -m pytest tests -x -q -m gpu -k "ef or EF or Fisher or fisher or updates or merged or golden or mc" > $OUT/ef_tests.txt 2>&1; tail -4 $OUT/ef_tests.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/ef_timing.txt
import time, torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
for dims, N in (([128, 256, 64, 10], 64), ([1024, 2688, 2688, 10], 8), ([1024, 2688, 2688, 10], 128)):
    m = nn.Sequential(nn.Linear(dims[0], dims[1]), nn.ReLU(), nn.Linear(dims[1], dims[2]), nn.ReLU(), nn.Linear(dims[2], dims[3])).to(dev)
    X, y = torch.rand(N, dims[0], device=dev), torch.randint(0, dims[3], (N,), device=dev)
    for cls in (C.GGNLinearOperator, C.EFLinearOperator):
        op = cls(m, nn.CrossEntropyLoss(), dict(m.named_parameters()), [(X, y)], check_deterministic=False)
        v = torch.rand(op.shape[1], device=dev)
        for _ in range(10): op @ v
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): op @ v
        torch.cuda.synchronize(); print(f"dims {dims} N={N} {cls.__name__}: {1e4*(time.perf_counter()-t0):.1f} us per product", flush=True)
PY
