"""Step-0 gradient accuracy of the DSFVT model (b = 2): l2 distance of every parameter gradient from an fp64 oracle run, for this
package and for the CPU fp32 oracle; prints the ratio, largest first.  LVT_NO_FLASH_ATTENTION=1 for the plane kernels."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")):
    sys.path.insert(0, p)
import torch, seeded
from oracle import lvt_oracle as O
from util_models import dsfvt_cfg
from lvt_amd.modeling import build_model
from lvt_amd.utils.events import EventStorage
seed = 47
model = build_model(dsfvt_cfg())
params = seeded.seeded_params(seeded.dsfvt_shapes(), seed)
model.model.load_state_dict(params, strict=False)
model.train()
ds = dict(blocks_e=((1, 16, 16),) * 8, blocks_d=((1, 16, 16),) * 8, stride=(16, 1, 1))
data = [O.prepare_slices(seeded.seeded_codes("traj.codes0.%d" % j, (16, 4, 16, 16), seed), (a, 0, 0), (16, 1, 1), (7, 1, 1), 1)
        for j, a in enumerate((2, 8))]
with EventStorage(0):
    loss = model(data, mode="supervised")["loss_cross_entropy"]
loss.backward()
ctx = torch.stack([d["context"] for d in data]); sl = torch.stack([d["slice"] for d in data])
si = torch.stack([d["slice_idx"] for d in data]); ig = torch.stack([d["ignore_mask"] for d in data])
g = {}
for dt in (torch.float32, torch.float64):
    p = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
    lo, _ = O.vt_supervised_loss(p, ctx, sl, si, ig, **ds)
    lo.backward()
    g[dt] = {k: v.grad.double() for k, v in p.items()}
named = dict(model.model.named_parameters())
rows = []
for k, r64 in g[torch.float64].items():
    if named[k].grad is None:
        continue
    n = r64.norm() + 1e-300
    em, ec = float((named[k].grad.double().cpu() - r64).norm() / n), float((g[torch.float32][k] - r64).norm() / n)
    rows.append((em / max(ec, 1e-12), k, em, ec))
rows.sort(reverse=True)
import statistics
print("flash_off=%s  median ratio %.2f  max ratio %.2f" % (os.environ.get("LVT_NO_FLASH_ATTENTION"), statistics.median(r[0] for r in rows), rows[0][0]))
for r in rows[:14]:
    print("  %6.2f  %-55s mine %.3e  cpu32 %.3e" % r)
