import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvt_amd.modeling.autoregressive.vt_attention as A
dev = "cuda:0"
A.PLANE_ATTENTION = os.environ.get("PLANES", "1") == "1"
for masked in (False, True):
    torch.manual_seed(0)
    layer = A.BlockLocalAttention((1, 16, 16), 128, 512, 8, masked=masked).to(dev)
    x = torch.randn(64 * 256, 512, device=dev)
    gy = torch.randn_like(x)
    for _ in range(30):
        xx = x.clone().requires_grad_(True)
        layer.forward_tokens(xx, layer.block_size).backward(gy)
    torch.cuda.synchronize()
