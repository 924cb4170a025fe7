import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests/golden")]
import torch, seeded
from conftest import Golden
from oracle import lvt_oracle as O
from util_models import dsfvt_cfg
from lvt_amd.modeling import build_model
from lvt_amd.utils.events import EventStorage
g = Golden("g12_dsfvt_loss")
model = build_model(dsfvt_cfg())
params = seeded.seeded_params(seeded.dsfvt_shapes(), int(g["seed"]))
model.model.load_state_dict(params, strict=False)
data = [O.prepare_slices(g["codes"][i], (int(g["a"][i]), 0, 0), (16, 1, 1), (7, 1, 1), 1) for i in range(2)]
model.train()
with EventStorage(0):
    loss = model(data, mode="supervised")["loss_cross_entropy"]
loss.backward()
mine = model.model.encoder.conv.weight.grad.cpu()[..., 0, 0]
p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
ctx = torch.stack([d["context"] for d in data]); sl = torch.stack([d["slice"] for d in data])
si = torch.stack([d["slice_idx"] for d in data]); ig = torch.stack([d["ignore_mask"] for d in data])
lo, _ = O.vt_supervised_loss(p, ctx, sl, si, ig, blocks_e=((1,16,16),)*8, blocks_d=((1,16,16),)*8, stride=(16,1,1))
lo.backward()
ref = p["encoder.conv.weight"].grad[..., 0, 0]
diff = (mine - ref).abs()
print("max ref", float(ref.abs().max()), "max diff", float(diff.max()), "n nonzero ref", int((ref != 0).sum()), "n nonzero mine", int((mine != 0).sum()))
idx = torch.nonzero(diff > 0.1 * diff.max())
print("num large diffs", idx.shape[0])
for o, cc, tau in idx[:20].tolist():
    print(o, cc, cc // 512, cc % 512, tau, float(mine[o, cc, tau]), float(ref[o, cc, tau]))
# count how many times each (c, code, tau) appears in context
cnt = torch.zeros(2048, 7)
for b in range(2):
    for c in range(4):
        for tau in range(7):
            v = ctx[b, c, tau].reshape(-1)
            v = v[v >= 0]
            cnt[:, tau] += torch.bincount(c * 512 + v, minlength=2048).float()
print("max count", float(cnt.max()))
for o, cc, tau in idx[:10].tolist():
    print("count at", cc, tau, float(cnt[cc, tau]))
