import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests/golden")]
import torch, seeded
from oracle import lvt_oracle as O
from util_models import dsfvt_cfg
from lvt_amd.modeling import build_model
from lvt_amd.utils.events import EventStorage
model = build_model(dsfvt_cfg())
params = seeded.seeded_params(seeded.dsfvt_shapes(), 4321)
model.model.load_state_dict(params, strict=False)
model.train()
for trial in range(4):
    data = [O.prepare_slices(seeded.seeded_codes("t%d.%d" % (trial, i), (16, 4, 16, 16), 5), (a, 0, 0), (16, 1, 1), (7, 1, 1), 1)
            for i, a in enumerate((3 + trial, 11 - trial))]
    model.model.zero_grad()
    with EventStorage(0):
        loss = model(data, mode="supervised")["loss_cross_entropy"]
    loss.backward()
    p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    ctx = torch.stack([d["context"] for d in data]); sl = torch.stack([d["slice"] for d in data])
    si = torch.stack([d["slice_idx"] for d in data]); ig = torch.stack([d["ignore_mask"] for d in data])
    # capture the gradient w.r.t. the encoder front output (tokens) in the oracle via a hook on zl's input
    lo, _ = O.vt_supervised_loss(p, ctx, sl, si, ig, blocks_e=((1,16,16),)*8, blocks_d=((1,16,16),)*8, stride=(16,1,1))
    lo.backward()
    for n in ("encoder.conv.weight", "encoder.linear_projector.weight", "encoder.conv.bias", "decoder.ch_embedder.1.weight"):
        a = dict(model.model.named_parameters())[n].grad.double().cpu(); b = p[n].grad.double()
        print(trial, n, "l2", float((a-b).norm()/b.norm()), "maxrel", float((a-b).abs().max()/b.abs().max()))
    mine = model.model.encoder.conv.weight.grad.cpu()[..., 0, 0]; ref = p["encoder.conv.weight"].grad[..., 0, 0]
    diff = (mine - ref).abs().sum(0)   # (2048, 7)
    big = torch.nonzero(diff > 0.2 * diff.max())
    # map back to pixels
    hits = {}
    for cc, tau in big.tolist():
        c, code = cc // 512, cc % 512
        m = (ctx[:, c, tau] == code)
        for b_, h, w in torch.nonzero(m).tolist():
            hits[(b_, h, w)] = hits.get((b_, h, w), 0) + 1
    top = sorted(hits.items(), key=lambda kv: -kv[1])[:3]
    print(trial, "entries", big.shape[0], "top pixels", top, "slice a", [int(d["slice_idx"]) for d in data])
