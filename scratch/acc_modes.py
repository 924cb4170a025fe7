"""Error of the engine vs an fp64 product in both math modes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import gemm as G, binding as L
dev = "cuda:0"
torch.manual_seed(0)
def probe(name, a, b):
    M, K = a.shape; N = b.shape[0]
    ref = a.double() @ b.double().t()
    scale = (a.double().abs() @ b.double().abs().t())          # sum |a||b| : the natural error scale
    out = {}
    for mode in ("f32", "bf16x3"):
        L.set_math_mode(mode)
        c = torch.empty(M, N, device=dev)
        G.gemm(a, b, c, M, N, K)
        e = (c.double() - ref).abs()
        out[mode] = (float((e / scale).max()), float((e / scale).pow(2).mean().sqrt()), float(e.max() / ref.abs().max()))
    t = a @ b.t()
    e = (t.double() - ref).abs()
    out["torch"] = (float((e / scale).max()), float((e / scale).pow(2).mean().sqrt()), float(e.max() / ref.abs().max()))
    print(name, {k: "max %.2e rms %.2e relmax %.2e" % v for k, v in out.items()})
for K in (64, 512, 4096, 32768):
    a = torch.randn(512, K, device=dev); b = torch.randn(512, K, device=dev)
    probe("normal K=%d" % K, a, b)
a = torch.randn(512, 4096, device=dev) * torch.exp(4 * torch.randn(512, 4096, device=dev))
b = torch.randn(512, 4096, device=dev) * torch.exp(4 * torch.randn(512, 4096, device=dev))
probe("lognormal K=4096", a, b)
a = torch.rand(512, 4096, device=dev); b = torch.rand(512, 4096, device=dev)
probe("uniform+ K=4096", a, b)
a = torch.randn(512, 4096, device=dev) * 1e-6; b = torch.randn(512, 4096, device=dev) * 1e-5
probe("tiny K=4096", a, b)
a = torch.softmax(torch.randn(512, 256, device=dev) * 3, -1); b = torch.randn(512, 256, device=dev) * 1e-4
probe("softmax x small K=256", a, b)
