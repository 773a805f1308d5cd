import torch, time
for n in (576, 1152, 2304, 4608):
    A = torch.randn(n, n, device="cuda"); A = A @ A.T / n + 1e-3 * torch.eye(n, device="cuda")
    torch.linalg.eigh(A); torch.cuda.synchronize()
    t0 = time.perf_counter(); L, Q = torch.linalg.eigh(A); torch.cuda.synchronize(); t = time.perf_counter() - t0
    err = ((Q * L) @ Q.T - A).abs().max().item() / A.abs().max().item()
    t1 = time.perf_counter(); Lc, Qc = torch.linalg.eigh(A.cpu().double()); tc = time.perf_counter() - t1
    print(f"n={n}: torch.linalg.eigh gpu {t*1e3:.1f} ms (recon err {err:.1e}); cpu f64 {tc*1e3:.0f} ms")
