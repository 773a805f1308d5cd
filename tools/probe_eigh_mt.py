import torch, time, threading
ns = [4608]*3 + [2304]*4 + [1152]*4 + [576]*4 + [512]*5
mats = []
for n in ns:
    A = torch.randn(n, n, device="cuda"); mats.append(A @ A.T / n + 1e-3 * torch.eye(n, device="cuda"))
def seq():
    return [torch.linalg.eigh(A) for A in mats]
seq(); torch.cuda.synchronize()
t0 = time.perf_counter(); r = seq(); torch.cuda.synchronize(); print(f"sequential: {(time.perf_counter()-t0)*1e3:.1f} ms")
def par(T):
    out = [None] * len(mats)
    main = torch.cuda.current_stream(); ev = main.record_event()
    def work(tid):
        s = torch.cuda.Stream()
        s.wait_event(ev)
        with torch.cuda.stream(s):
            for i in range(tid, len(mats), T):
                out[i] = torch.linalg.eigh(mats[i])
        main_ev = s.record_event(); evs[tid] = main_ev
    evs = [None] * T
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    for e in evs: main.wait_event(e)
    return out
for T in (2, 4, 8):
    par(T); torch.cuda.synchronize()
    t0 = time.perf_counter(); o = par(T); torch.cuda.synchronize(); print(f"{T} threads/streams: {(time.perf_counter()-t0)*1e3:.1f} ms")
    err = max(((q[1] * q[0]) @ q[1].T - A).abs().max().item() for q, A in zip(o, mats))
    print("  max recon err", err)
