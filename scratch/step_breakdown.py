import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from lvt_amd.hip import binding as L
cfg, model = bench.build_vqvae("cuda:0", 1)
opts, _ = model.configure_optimizers_and_checkpointers()
clips = torch.rand(32, 16, 3, 64, 64).cuda()
data = [{"image_sequence": clips[i]} for i in range(32)]
for i in range(3): bench.vqvae_step(model, opts, data, i)
torch.cuda.synchronize()
L.TIMER = L.KernelTimer()
for i in range(5): bench.vqvae_step(model, opts, data, i)
s = L.TIMER.summary(); L.TIMER = None
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-18s launches/step %3d  ms/step %7.3f  avg_us %8.1f  TF %6.1f" % (k, v["launches"] // 5, v["ms"] / 5, v["ms"] / v["launches"] * 1e3, v["flops"] / (v["ms"] * 1e-3) / 1e12))
