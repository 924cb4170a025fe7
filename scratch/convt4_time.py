"""lvt_convt4_fwd at the bench shape (512 frames x 32x32 x 128 -> 64x64 x 3): us per launch with the library named by LVT_HIP_LIB"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from lvt_amd.hip import binding as L, gemm as G
dev = "cuda:0"
x = torch.randn(512, 32, 32, 128, device=dev); w = torch.randn(128, 3, 4, 4, device=dev) * 0.05; b = torch.randn(3, device=dev)
def run(): return G.convT4_fwd(x.view(512, 1, 32, 32, 128), w, b, True)
for _ in range(3): y = run()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
ref = torch.tanh(torch.nn.functional.conv_transpose2d(x[:8].permute(0, 3, 1, 2), w, b, stride=2, padding=1)).permute(0, 2, 3, 1)
print(os.path.basename(os.environ.get("LVT_HIP_LIB", "default")), "%.1f us" % (a.elapsed_time(e) / 20 * 1e3), "max err %.2e" % float((y[:8, 0, ..., :3] - ref).abs().max()))
