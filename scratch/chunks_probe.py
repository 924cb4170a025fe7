"""top errors of test_dsfvt_train_step_64_slices_equals_mean_of_chunks (full batch vs mean of four chunks)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import test_gpu_fullsize as T
from conftest import rel_err
from lvt_amd.data.dataset_mapper import prepare_slices_batch
from lvt_amd.modeling import build_model
from lvt_amd.hip import binding as L
L.set_math_mode("f16x2")
DEV = T.DEV
cfg = T.dsfvt_cfg(DEV)
torch.manual_seed(9)
model = build_model(cfg); model.train()
with torch.no_grad():
    for n, p in model.model.named_parameters():
        if n.endswith("_bank"): p.normal_(0, 0.2)
v = cfg.MODEL.AUTOREGRESSIVE.VT
g = torch.Generator().manual_seed(23)
codes = torch.randint(0, v.NV, (64, 16, v.NC, 16, 16), generator=g).to(DEV)
abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (64,), generator=g)]
def run(lo, hi):
    for p in model.parameters(): p.grad = None
    ctx, sl, sidx, ign = prepare_slices_batch(codes[lo:hi], abcs[lo:hi], v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)
    loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
    loss.backward()
    return float(loss), T._grads([model.model])
lf, gf = run(0, 64)
parts = [run(16 * c, 16 * c + 16) for c in range(4)]
errs = []
for k, a in gf.items():
    acc = sum(p[1][k].double() for p in parts) / 4
    errs.append((rel_err(a, acc), k, float(a.abs().max())))
errs.sort(reverse=True)
print(os.environ.get("LVT_FA_TWOPASS"), [(("%.2e" % e), k, "%.2e" % m) for e, k, m in errs[:8]])
