import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import tx
dev = "cuda:0"
B, H, S, da = 64, 8, 256, 128
q, k, v = (torch.randn(B * S, H * da, device=dev) for _ in range(3))
banks = [torch.randn(H, 2 * n - 1, device=dev) * 0.3 for n in (1, 16, 16)]
def run(): tx.attn_fwd(q, k, v, B, H, S, da, math.sqrt(da), banks[0], banks[1], banks[2], (1, 16, 16), True)
for _ in range(3): run()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20
print("attn_fwd b64 h8: %.1f us  %.1f TF algorithmic" % (t * 1e3, 4.0 * B * H * S * S * da / t / 1e9))
