import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from lvt_amd.hip import binding as L, gemm as G, ew
from util_models import dsfvt_cfg
DEV = "cuda:0"
rec = []
orig = G.gemm_p2
def spy(A, Bimg, C_out, M, N, K, **kw):
    r = orig(A, Bimg, C_out, M, N, K, **kw)
    rec.append((C_out.clone(), (A.data if isinstance(A, G.P2Image) else A).clone(), Bimg.data.clone(), float(A.amax) if isinstance(A, G.P2Image) else None, float(Bimg.amax)))
    return r
G.gemm_p2 = spy
import lvt_amd.modeling.autoregressive.vt_attention as VA
def run():
    from lvt_amd.data.dataset_mapper import prepare_slices_batch
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    cfg = dsfvt_cfg(DEV); cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
    torch.manual_seed(13)
    model = build_model(cfg); model.train()
    v = cfg.MODEL.AUTOREGRESSIVE.VT
    g = torch.Generator().manual_seed(7)
    codes = torch.randint(0, v.NV, (8, 16, v.NC, 16, 16), generator=g).to(DEV)
    abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (8,), generator=g)]
    ctx, sl, sidx, ign = prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)
    rec.clear()
    with EventStorage(0):
        loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
    return float(loss), list(rec)
l0, r0 = run()
junk = [torch.full((1 << 24,), float("nan"), device=DEV) for _ in range(8)]; del junk
l1, r1 = run()
print("losses", l0, l1, "calls", len(r0), len(r1))
for i, (a, b) in enumerate(zip(r0, r1)):
    same = [torch.equal(x, y) if torch.is_tensor(x) else x == y for x, y in zip(a, b)]
    if not all(same):
        print("call", i, "C/A/B/a_amax/b_amax equal:", same, "shape", tuple(a[0].shape), "nan in C:", bool(torch.isnan(a[0]).any()), bool(torch.isnan(b[0]).any()))
        d = (a[0] != b[0])
        print("   differing C elements:", int(d.sum()), "rows:", d.reshape(-1, d.shape[-1]).any(1).nonzero().flatten()[:10].tolist())
        break
else:
    print("all recorded calls identical")
# replay call 0 from its recorded images: batched vs block by block, several times
C0, Ai, Bi, aam, bam = r0[0]
M, d, na, da = 2048, 512, 8, 128
hd = na * da
aA, aB = torch.tensor([aam], device=DEV), torch.tensor([bam], device=DEV)
outs = []
for rep in range(4):
    C = torch.full((3, M, hd), float("nan"), device=DEV)
    orig(G.P2Image(Ai, aA), G.P2Image(Bi, aB), C, M, da, d, lda=d, ldb=d, ldc=hd, batch_outer=3, batch_inner=na, sB=(na * da * d, da * d), sC=(M * hd, da))
    outs.append(C)
print("batched replays equal to each other:", [torch.equal(outs[0], o) for o in outs[1:]], "to run0:", torch.equal(outs[0], C0), "to run1:", torch.equal(outs[0], r1[0][0]))
ref = torch.empty(3, M, hd, device=DEV)
for z in range(24):
    blk = torch.empty(M, da, device=DEV)
    orig(G.P2Image(Ai, aA), G.P2Image(Bi[z * da:(z + 1) * da], aB), blk, M, da, d)
    ref[z // na, :, (z % na) * da:(z % na + 1) * da] = blk
print("block-by-block == batched replay:", torch.equal(ref, outs[0]), " == run0:", torch.equal(ref, C0), " == run1:", torch.equal(ref, r1[0][0]))
for name, X in (("run0", C0), ("run1", r1[0][0]), ("replay", outs[0])):
    dd = (X != ref)
    print(name, "differs from block-by-block in", int(dd.sum()), "elements; per projection:", [int(dd[p_].sum()) for p_ in range(3)], "heads:", [int(dd[:, :, h_ * da:(h_ + 1) * da].sum()) for h_ in range(na)])
