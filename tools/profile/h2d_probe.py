import time, torch, numpy as np
torch.cuda.set_device(0)
arrs = [np.random.rand(16, 3, 64, 64).astype(np.float32) for _ in range(32)]
host = torch.empty((32, 16, 3, 64, 64), pin_memory=True)
pag = torch.empty((32, 16, 3, 64, 64))
def T(name, fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print("%-40s %.2f ms" % (name, (time.perf_counter() - t) / n * 1e3))
T("32 copies into pinned", lambda: [host[i].copy_(torch.as_tensor(a)) for i, a in enumerate(arrs)])
T("32 copies into pageable", lambda: [pag[i].copy_(torch.as_tensor(a)) for i, a in enumerate(arrs)])
T("np.stack", lambda: np.stack(arrs))
T("np.stack into pinned (out=)", lambda: np.stack(arrs, out=host.numpy()))
T("H2D from pinned (25 MB)", lambda: host.to("cuda:0", non_blocking=True))
T("H2D from pageable", lambda: pag.to("cuda:0"))
T("torch.empty pinned 25 MB", lambda: torch.empty((32, 16, 3, 64, 64), pin_memory=True), 3)
print(torch.get_num_threads())
