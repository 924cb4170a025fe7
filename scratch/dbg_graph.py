import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests/golden")]
import torch, seeded
from util_models import dsfvt_cfg
from lvt_amd.modeling import build_model
from lvt_amd.modeling.autoregressive.incremental import GraphedSliceSampler, IncrementalDecoder
cfg = dsfvt_cfg(); model = build_model(cfg).eval()
B = 16
ctx = torch.randint(0, 512, (B, 4, 7, 16, 16), device="cuda:0"); sl = torch.randint(0, 512, (B, 4, 1, 16, 16), device="cuda:0")
si = torch.full((B,), 6, dtype=torch.long, device="cuda:0")
with torch.no_grad():
    zl = model.model.encoder.forward_tokens(ctx, si)
    s = GraphedSliceSampler(model.model, B, (1, 16, 16), 1.0)
    s.begin_slice(zl, sl)
    s._body(3, True); torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s._body(3, True)
        print("capture with sampling OK")
        t = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); print("replay ms", (time.perf_counter() - t) / 50 * 1e3)
    except Exception:
        traceback.print_exc()
    torch.cuda.synchronize()
    try:
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            s._body(4, False)
        print("capture decoder-only OK")
        t = time.perf_counter()
        for _ in range(50): g2.replay()
        torch.cuda.synchronize(); print("replay ms", (time.perf_counter() - t) / 50 * 1e3)
    except Exception:
        traceback.print_exc()
    t = time.perf_counter()
    for _ in range(20): s._body(5, True)
    torch.cuda.synchronize(); print("eager ms", (time.perf_counter() - t) / 20 * 1e3)
