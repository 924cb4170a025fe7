import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import vq
dev = "cuda:0"
COARSE = os.environ.get("LVT_VQ_COARSE") == "1"
FR = int(os.environ.get("FRAMES", "512"))
torch.manual_seed(0)
for scale, name in ((1.0 / 512, "init-like codebook U(-1/512, 1/512), z ~ N(0,1)"), (1.0, "codebook ~ N(0,1), z ~ N(0,1)")):
    z = torch.randn(FR * 256, 256, device=dev)
    cb = (torch.rand(4, 512, 64, device=dev) * 2 - 1) * scale if scale < 1 else torch.randn(4, 512, 64, device=dev)
    idx = vq.nearest(z, cb, 256, coarse=COARSE)
    # fp64 check
    bad = 0
    for g in range(4):
        x = z[:, g * 64:(g + 1) * 64].double(); e = cb[g].double()
        best = torch.empty(x.shape[0], dtype=torch.int64, device=dev)
        for s0 in range(0, x.shape[0], 32768):
            d = (e ** 2).sum(1)[None] + (x[s0:s0 + 32768] ** 2).sum(1, keepdim=True) - 2 * x[s0:s0 + 32768] @ e.t()
            best[s0:s0 + 32768] = d.argmin(1)
        mine = idx.view(FR, 4, 256)[:, g].reshape(-1)
        bad += int((mine != best).sum())
    for _ in range(3): vq.nearest(z, cb, 256, coarse=COARSE)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): vq.nearest(z, cb, 256, coarse=COARSE)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print("%-50s %7.1f us  (%.2f TB/s of 270,336 B/frame)  rows differing from the fp64 argmin: %d of %d" % (name, us, 270336 * FR / us / 1e6, bad, 4 * z.shape[0]))
