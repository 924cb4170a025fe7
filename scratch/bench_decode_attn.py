"""Decode attention: time and achieved K/V-cache bandwidth per query position and batch (8 layers worth of distinct
caches are cycled so that nothing stays in the 256 MB Infinity Cache)."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import tx
dev = "cuda:0"
H, da, S = 8, 128, 256
hd = H * da
for B in (64, 256):
    nl = 8
    K = [torch.randn(B, S, hd, device=dev) for _ in range(nl)]
    V = [torch.randn(B, S, hd, device=dev) for _ in range(nl)]
    q = torch.randn(B, hd, device=dev)
    dt, dh, dw = torch.zeros(H, 1, device=dev), torch.zeros(H, 31, device=dev), torch.zeros(H, 31, device=dev)
    for qi in (31, 127, 255):
        def run():
            for l in range(nl):
                tx.attn_decode(q, K[l], V[l], H, qi, math.sqrt(da), dt, dh, dw, (1, 16, 16))
        run(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) / (10 * nl) * 1e3
        byts = B * (qi + 1) * hd * 4 * 2
        print("B %3d qi %3d: %6.1f us  %5.2f TB/s of K/V reads" % (B, qi, t, byts / t / 1e6))
