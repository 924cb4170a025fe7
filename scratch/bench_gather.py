import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import vq
dev = "cuda:0"
n, num, P, KC, D = 512, 4, 256, 512, 64
idx = torch.randint(0, KC, (n, num, P), device=dev)
E = torch.randn(num, KC, D, device=dev)
out = vq.gather(idx, E)
ref = torch.stack([E[g][idx[:, g]] for g in range(num)], 2).reshape(n * P, num * D)   # (n,P,num,D)
print("exact:", bool((out.reshape(n * P, num * D) == ref).all()), tuple(out.shape))
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): vq.gather(idx, E)
e.record(); torch.cuda.synchronize()
t = a.elapsed_time(e) / 20
print("vq_gather %.1f us  (%.2f TB/s written)" % (t * 1e3, n * P * num * D * 4 / t / 1e9))
