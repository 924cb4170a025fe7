"""Plane-operand pipelined attention (csrc/attention_pipe.hip) against the round-2 path (attn_fwd + 4 GEMMs + softmax-bwd)
on one BlockLocalAttention layer: outputs, every gradient, and the time of both."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvt_amd.modeling.autoregressive.vt_attention as A
dev = "cuda:0"


def run(layer, x, gy, planes):
    A.PLANE_ATTENTION = planes
    for p in layer.parameters():
        p.grad = None
    xx = x.clone().requires_grad_(True)
    y = layer.forward_tokens(xx, layer.block_size)
    y.backward(gy)
    return [y.detach(), xx.grad] + [p.grad.clone() for p in layer.parameters()]


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


for block in ((1, 16, 16), (4, 8, 8)):
    for masked in (False, True):
        torch.manual_seed(0)
        layer = A.BlockLocalAttention(block, 128, 512, 8, masked=masked).to(dev)
        with torch.no_grad():
            layer.dt_bank.normal_(0, 0.3); layer.dh_bank.normal_(0, 0.3); layer.dw_bank.normal_(0, 0.3)
        b = 8
        x = torch.randn(b * 256, 512, device=dev)
        gy = torch.randn_like(x)
        new = run(layer, x, gy, True)
        old = run(layer, x, gy, False)
        names = ["y", "dx"] + [n for n, _ in layer.named_parameters()]
        # (the gradient of a one-entry bank is sum_ij g_ij == 0 up to rounding: judge it against the scale of the other banks)
        scale = {n: (float(old[names.index("dh_bank")].abs().max()) if n.endswith("_bank") else None) for n in names}
        rel = lambda a, c, n=None: float((a - c).abs().max() / ((scale.get(n) or float(c.abs().max())) + 1e-30))
        worst = max(rel(a, c, n) for n, a, c in zip(names, new, old))
        print("block", block, "masked", masked, "worst rel diff %.2e" % worst,
              {n: "%.1e" % rel(a, c, n) for n, a, c in zip(names, new, old) if rel(a, c, n) > 1e-6})
        assert worst < 2e-5, worst

# timing at the bench shape
torch.manual_seed(0)
for masked in (False, True):
    layer = A.BlockLocalAttention((1, 16, 16), 128, 512, 8, masked=masked).to(dev)
    x = torch.randn(64 * 256, 512, device=dev)
    gy = torch.randn_like(x)
    for planes in (True, False):
        for _ in range(3):
            run(layer, x, gy, planes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run(layer, x, gy, planes)
        torch.cuda.synchronize()
        print("masked" if masked else "full  ", "planes" if planes else "old   ", "%.3f ms per layer fwd+bwd" % ((time.perf_counter() - t0) * 100))
