"""Where does a forward attention workgroup spend its time?  Needs the timestamp build (LVT_HIP_LIB=scratch/variants/lib_ts.so:
wave 0 of every workgroup stamps s_memrealtime (100 MHz) at its phase boundaries into the tail of the P buffer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lvt_amd.hip import binding as L
dev = "cuda:0"
B, H, S, da = 64, 8, 256, 128
M, hd = B * S, H * da
torch.manual_seed(0)
q = torch.randn(3, M, hd, device=dev)
planes = torch.empty(3, 3, M, hd, dtype=torch.bfloat16, device=dev)
r = q.clone()
for pl in range(3):
    planes[:, pl] = r.to(torch.bfloat16)
    r = r - planes[:, pl].float()
dt = torch.zeros(H, 1, device=dev); dh = torch.randn(H, 31, device=dev) * 0.1; dw = torch.randn(H, 31, device=dev) * 0.1
nwg = B * H * 2
P = torch.empty(B * H * S * S + nwg * 16 * 2, dtype=torch.float32, device=dev)
o = torch.empty(M, hd, dtype=torch.float32, device=dev)
lib = L.lib()
for it in range(3):
    P[B * H * S * S:].zero_()
    torch.cuda.synchronize()
    L.check(lib.lvt_attn_fwd_planes(L.ptr(planes), M * hd, 3 * M * hd, B, H, S, da, float(da) ** 0.5, L.ptr(dt), L.ptr(dh), L.ptr(dw),
                                    1, 16, 16, 0, -1e4, L.ptr(P), L.ptr(o), L.stream_ptr()), "fwd")
    torch.cuda.synchronize()
ts = P[B * H * S * S:].view(torch.int64).view(nwg, 16).cpu().numpy()
t = ts[:, :12].astype(np.float64) / 100.0        # us
t0 = t[:, 0].min()
names = ["prologue (loads of K0, K1, Q, tables; park K0; barrier)", "K chunk 0", "K chunk 1", "K chunk 2", "K chunk 3",
         "last softmax + normalisation", "V chunk 0", "V chunk 1", "V chunk 2", "V chunk 3", "O stores"]
d = np.diff(t, axis=1)
print("workgroups %d; kernel span %.1f us (first start .. last end)" % (nwg, t[:, 11].max() - t0))
print("workgroup life: mean %.2f us, min %.2f, max %.2f" % ((t[:, 11] - t[:, 0]).mean(), (t[:, 11] - t[:, 0]).min(), (t[:, 11] - t[:, 0]).max()))
for k, nme in enumerate(names):
    print("  %-58s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (nme, d[:, k].mean(), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
start = np.sort(t[:, 0] - t0)
print("start times: first wave of %d workgroups starts within %.2f us; then quartiles %s" % (256, start[255], np.percentile(start, [25, 50, 75, 100]).round(1)))
hw = ts[:, 15]
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7
print("distinct (se, cu) pairs seen:", len(set(zip(se.tolist(), cu.tolist()))))
