import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = "cuda:0"
vq = bench.VqvaeLeg(dev, 1, 0, 0, 32, 2)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    vq.step(i)
m = vq.model
with torch.no_grad():
    x_cl, _ = m._preprocess_cl(vq.batches[0])
    z = m.encoder.forward_cl(x_cl).view(-1, 4, 64)
    sd = m.codebook.state_dict()
    for g in range(4):
        e = sd["ve.%d.embedding.weight" % g]
        x = z[:, g]
        xn = x.norm(dim=1); en = e.norm(dim=1)
        sc = x.bfloat16().float() @ e.bfloat16().float().t() - 0.5 * (e ** 2).sum(1)[None]
        mx = sc.max(1, keepdim=True).values
        band = 2 * 1.002 / 256 * xn[:, None] * en.max() + 1e-5 * (xn[:, None] ** 2 + en.max() ** 2 + 2 * xn[:, None] * en.max())
        nc = (sc >= mx - band).sum(1)
        spread = sc.std(1)
        print("group", g, "|x| mean %.3f  |e| mean %.4f max %.4f  score std %.3e  band %.3e  candidates/row mean %.1f max %d  rows>16: %d of %d"
              % (float(xn.mean()), float(en.mean()), float(en.max()), float(spread.mean()), float(band.mean()), float(nc.float().mean()), int(nc.max()),
                 int((nc > 16).sum()), nc.numel()))
