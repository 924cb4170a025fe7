"""GPU parity of the other shipped shapes against fixtures captured from the reference:
DSSVT (spatial subscaling; block-split attention at the 16-frame test length), DSTSVT (spatio-temporal
subscaling, (5,3,3) context conv), class-conditional DSFVT, K-DVQVAE (4 residual blocks per side).
Tolerances as in test_gpu_vt / test_gpu_vqvae: loss 2e-5 relative, logits 1e-4 of the tensor max, gradient
norms 2e-3, selected gradient entries at the roundoff class bound 1e-2 (two fp32 evaluation orders)."""
import os

import pytest
import torch
import torch.nn.functional as F

import seeded
from conftest import rel_err
from oracle import lvt_oracle as O
from util_models import MEAN, ROOT, STD, margin_ok

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VARIANTS = {
    "g15_dssvt": dict(cfg="configs/vt/DSSVT.yaml", block=(4, 8, 8), kernel=(1, 3, 3), stride=(1, 2, 2), n_slices=4),
    "g16_dstsvt": dict(cfg="configs/vt/DSTSVT.yaml", block=(4, 8, 8), kernel=(5, 3, 3), stride=(4, 2, 2), n_slices=16),
    "g17_dsfvt_class": dict(cfg="configs/vt/DSFVT.yaml", block=(1, 16, 16), kernel=(7, 1, 1), stride=(16, 1, 1),
                            n_slices=16, class_num=10),
}


def _build(tag, g, evaluators=None):
    from lvt_amd.config import get_cfg
    from lvt_amd.modeling import build_model
    v = VARIANTS[tag]
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, v["cfg"]))
    cfg.MODEL.DEVICE = "cuda"
    cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
    cfg.MODEL.AUTOREGRESSIVE.VT.CLASS_NUM = v.get("class_num", 0)
    if evaluators:
        cfg.TEST.EVALUATORS = evaluators
    model = build_model(cfg)
    params = seeded.seeded_params(seeded.dsfvt_shapes(block=v["block"], kernel=v["kernel"], n_slices=v["n_slices"],
                                                      class_num=v.get("class_num", 0)), int(g["seed"]))
    missing, unexpected = model.model.load_state_dict(params, strict=False)
    assert not unexpected and not (set(missing) & {n for n, _ in model.model.named_parameters()})
    return model, params, v


@pytest.mark.parametrize("tag", sorted(VARIANTS))
def test_variant_supervised_loss_and_grads(golden, tag):
    from lvt_amd.data.dataset_mapper import prepare_slices
    from lvt_amd.utils.events import EventStorage
    g = golden(tag)
    model, params, v = _build(tag, g)
    data = []
    for i in range(g["codes"].shape[0]):
        d = prepare_slices(g["codes"][i].numpy(), tuple(int(x) for x in g["abc"][i]), v["stride"], v["kernel"], 1, -1)
        if "class_num" in v:
            d["class"] = int(g["classes"][i])
        data.append(d)
    assert torch.equal(torch.stack([torch.as_tensor(d["context"]) for d in data]), g["context"])
    model.train()
    model.model.zero_grad()
    with EventStorage(0):
        loss = model(data, mode="supervised")["loss_cross_entropy"]
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-5 * float(g["loss"])
    named = dict(model.model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    norms = torch.tensor([float(named[n].grad.norm()) for n in names], dtype=torch.float64)
    ref = g["grad_norms"].double()
    worst = (norms - ref).abs() / (ref + 1e-6)
    assert float(worst.max()) < 2e-3, (names[int(worst.argmax())], float(worst.max()))
    assert rel_err(named["encoder.conv.weight"].grad[:2], g["grad_enc_conv_rows"]) < 1e-2
    assert rel_err(named["encoder.slice_embedding.weight"].grad, g["grad_slice_emb"]) < 1e-2
    assert rel_err(named["encoder.linear_projector.weight"].grad[:4, :, 0, 0, 0], g["grad_enc_proj_rows"]) < 1e-2
    if v["block"][0] > 1:
        assert rel_err(named["decoder.block_local_attention.3.dt_bank"].grad, g["grad_dec3_dt"]) < 1e-2
    if "class_num" in v:
        assert rel_err(named["encoder.class_embedding.weight"].grad, g["grad_class_emb"]) < 1e-2
    # forward activations through the reference-layout module contract
    ctx = g["context"].to(DEV)
    sl = torch.stack([torch.as_tensor(d["slice"]) for d in data]).to(DEV)
    si = g["slice_idx"].to(DEV)
    cls = g["classes"].long().to(DEV) if "class_num" in v else None
    with torch.no_grad():
        zl = model.model.encoder(ctx, si, class_idx=cls)
        pred = model.model(ctx, sl, si, mode="logits", class_idx=cls)
    assert rel_err(zl[:, ::16, :, ::3, ::3], g["zl_slice"]) < 5e-5
    assert rel_err(pred[0][:, ::8, :, ::3, ::3], g["logits0_slice"]) < 1e-4
    assert rel_err(pred[3][:, ::8, :, ::3, ::3], g["logits3_slice"]) < 1e-4


def test_dssvt_block_split_whole_video_logits(golden):
    """16-frame DSSVT evaluation: slices of (16,8,8) tokens, attention inside (4,8,8) blocks."""
    g = golden("g15_dssvt")
    model, _, _ = _build("g15_dssvt", g, evaluators="BitsEvaluator")
    model.eval()
    with torch.no_grad():
        out = model([{"image_sequence": g["eval_video"]}], mode="inference")[0]
    lg = out["logits"].cpu()
    assert tuple(lg.shape) == (4, 512, 16, 16, 16)
    assert torch.equal(out["ignore_mask"].cpu(), g["eval_ignore_mask"])
    assert rel_err(lg[:, ::64, ::3, ::5, ::5], g["eval_logits_slice"]) < 1e-4
    nll = F.cross_entropy(lg.permute(1, 0, 2, 3, 4)[None], g["eval_video"].transpose(0, 1)[None], reduction="none")[0]
    assert rel_err(nll, g["eval_nll"]) < 1e-4


def test_block_split_module_contract_matches_oracle():
    """BlockLocalAttention.forward on a volume larger than its block, with blocks along every axis."""
    from lvt_amd.modeling.autoregressive.vt_attention import BlockLocalAttention
    torch.manual_seed(3)
    layer = BlockLocalAttention((4, 8, 8), 128, 512, 8, masked=True).to(DEV)      # 256-token blocks
    with torch.no_grad():
        for n in ("dt_bank", "dh_bank", "dw_bank"):
            getattr(layer, n).copy_(0.3 * torch.randn_like(getattr(layer, n)))
    p = {"l." + k: v.detach().cpu() for k, v in layer.named_parameters()}
    x = torch.randn(1, 512, 8, 16, 16)                                             # 2 x 2 x 2 blocks
    xg = x.to(DEV).requires_grad_(True)
    y = layer(xg)
    gy = torch.randn_like(x)
    y.backward(gy.to(DEV))
    xr = x.clone().requires_grad_(True)
    yr = O.block_local_attention(p, "l.", xr, (4, 8, 8), masked=True)
    yr.backward(gy)
    assert rel_err(y, yr) < 2e-5
    assert rel_err(xg.grad, xr.grad) < 5e-5


def test_dstsvt_incremental_decoder_rows(golden):
    """K/V-cache decoding on a (4,8,8) slice (causal conv and bias banks with a time axis) == full pass rows."""
    from lvt_amd.data.dataset_mapper import prepare_slices
    from lvt_amd.modeling.autoregressive.incremental import IncrementalDecoder
    g = golden("g16_dstsvt")
    model, _, v = _build("g16_dstsvt", g)
    model.eval()
    data = [prepare_slices(g["codes"][i].numpy(), tuple(int(x) for x in g["abc"][i]), v["stride"], v["kernel"], 1, -1)
            for i in range(2)]
    ctx = g["context"].to(DEV)
    sl = torch.stack([torch.as_tensor(d["slice"]) for d in data]).to(DEV)
    si = g["slice_idx"].to(DEV)
    with torch.no_grad():
        zl = model.model.encoder.forward_tokens(ctx, si)
        full = model.model.decoder.forward_tokens(sl, zl).view(2, 256, 512)
        dec = IncrementalDecoder(model.model.decoder, zl, 2, (4, 8, 8))
        worst = 0.0
        for i in range(256):
            y = dec.step(sl, i)
            if i in (0, 1, 7, 8, 63, 64, 65, 130, 255):
                worst = max(worst, rel_err(y, full[:, i]))
    assert worst < 2e-5, worst


def test_kdvqvae_loss_grads_and_indices(golden):
    from lvt_amd.config import get_cfg
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    g = golden("g18_kdvqvae")
    seed = int(g["seed"])
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs/vqvae/K-DVQVAE.yaml"))
    cfg.MODEL.DEVICE = "cuda"
    cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
    model = build_model(cfg)
    es, ds = seeded.vqvae_shapes(4)
    model.encoder.load_state_dict(seeded.seeded_params(es, seed, "enc."))
    model.generator.load_state_dict(seeded.seeded_params(ds, seed, "dec."))
    st0 = seeded.seeded_codebook_state(seed, scale=0.6)
    model.codebook.load_state_dict(st0)
    model.train()
    with EventStorage(0):
        losses = model([{"image": g["x"][i].numpy()} for i in range(2)], mode="supervised")
    sum(losses.values()).backward()
    assert abs(float(losses["loss_reconstruction"].detach()) - float(g["loss_reconstruction"])) < 2e-5 * float(g["loss_reconstruction"])
    assert abs(float(losses["loss_commitment"].detach()) - float(g["loss_commitment"])) < 2e-4 * float(g["loss_commitment"])
    for sub, pre in ((model.encoder, "enc"), (model.generator, "dec")):
        named = dict(sub.named_parameters())
        names = [str(n) for n in g[pre + "_grad_names"]]
        norms = torch.tensor([float(named[n].grad.norm()) for n in names], dtype=torch.float64)
        ref = g[pre + "_grad_norms"].double()
        worst = (norms - ref).abs() / (ref + 1e-9)
        assert float(worst.max()) < 5e-3, (pre, names[int(worst.argmax())], float(worst.max()))
    assert rel_err(dict(model.generator.named_parameters())["layers.8.weight"].grad[:2], g["grad_dec_l8_rows"]) < 1e-2
    with torch.no_grad():
        z_e = model.encoder(model.normalizer(g["x"].to(DEV)))
    assert rel_err(z_e[:, ::8, ::2, ::2], g["z_e_slice"]) < 2e-5
    for i in range(4):
        assert rel_err(model.codebook.state_dict()["ve.%d.running_size" % i], g["new_ve.%d.running_size" % i]) < 1e-5


def test_g21_share_p_channel_predictor(golden):
    """SHARE_P = True -- the reference's CONFIG DEFAULT (vidgen/config/defaults.py:50): one output layer P for all channels
    (videotransformer.py:121-123,150-151).  Logits of every channel and the gradients (the shared P's is the sum over the
    channels) against the reference's own ChannelPredictor(share_p=True), fixture G21; state_dict keys as the reference's."""
    import seeded
    from lvt_amd.modeling.autoregressive.videotransformer import ChannelPredictor
    g = golden("g21_share_p")
    d, nc, nv, de = [int(x) for x in g["dims"]]
    cp = ChannelPredictor(d, nc, nv, de, share_p=True, share_embeddings=False)
    shapes = {k: tuple(v.shape) for k, v in cp.state_dict().items()}
    assert "P.weight" in shapes and "P.bias" in shapes and not any(k.startswith("P.0") for k in shapes)
    cp.load_state_dict(seeded.seeded_params(shapes, int(g["seed"]), "g21."))
    cp = cp.to(DEV)
    yl = g["yl"].to(DEV).requires_grad_(True)
    pred = cp(g["slice"].to(DEV), yl, mode="logits")
    sum((o * g["gy_%d" % k].to(DEV)).sum() for k, o in enumerate(pred)).backward()
    for k in range(nc):
        assert rel_err(pred[k], g["logits_%d" % k]) < 2e-5
    assert rel_err(cp.P.weight.grad, g["grad_P_weight"]) < 1e-4
    assert rel_err(cp.P.bias.grad, g["grad_P_bias"]) < 1e-4
    assert rel_err(cp.U[2].weight.grad, g["grad_U2_weight"]) < 1e-4
    assert rel_err(cp.U[0].bias.grad, g["grad_U0_bias"]) < 1e-4
    assert rel_err(cp.layer_norm.weight.grad, g["grad_ln_w"]) < 1e-4
    assert rel_err(yl.grad, g["grad_yl"]) < 1e-4
    # incremental sampling goes through the same shared layer
    with torch.no_grad():
        codes, probs = cp.sample_from_rows(torch.randn(4, d, device=DEV), forced_codes=torch.zeros(4, nc, dtype=torch.int64),
                                           return_probs=True)
    assert tuple(probs.shape) == (4, nc, nv) and torch.isfinite(probs).all()


def test_g24_share_embeddings_channel_predictor(golden):
    """SHARE_EMBEDDINGS (videotransformer.py:124-125,152-154,174-176): ONE layer P: d -> de, the decoder's channel embedding
    table E_k as the output matrix.  Logits and gradients (P, the tied tables, U_2, the input) against the reference's own
    ChannelPredictor(share_embeddings=True), fixture G24; the sampling path goes through the same tables."""
    import seeded
    from lvt_amd.modeling.autoregressive.videotransformer import ChannelPredictor
    g = golden("g24_share_embeddings")
    d, nc, nv, de = [int(x) for x in g["dims"]]
    cp = ChannelPredictor(d, nc, nv, de, share_p=False, share_embeddings=True)
    shapes = {k: tuple(v.shape) for k, v in cp.state_dict().items()}
    assert shapes["P.weight"] == (de, d) and not any(k.startswith("P.0") or "emb" in k for k in shapes)
    cp.load_state_dict(seeded.seeded_params(shapes, int(g["seed"]), "g24."))
    emb = torch.nn.ModuleList([torch.nn.Embedding(nv, de) for _ in range(nc)])
    emb.load_state_dict(seeded.seeded_params({"%d.weight" % k: (nv, de) for k in range(nc)}, int(g["seed"]), "g24.emb."))
    cp, emb = cp.to(DEV), emb.to(DEV)
    yl = g["yl"].to(DEV).requires_grad_(True)
    pred = cp(g["slice"].to(DEV), yl, mode="logits", ch_embedder=emb)
    sum((o * g["gy_%d" % k].to(DEV)).sum() for k, o in enumerate(pred)).backward()
    for k in range(nc):
        assert rel_err(pred[k], g["logits_%d" % k]) < 2e-5
        assert rel_err(emb[k].weight.grad, g["grad_emb_%d" % k]) < 1e-4
    assert rel_err(cp.P.weight.grad, g["grad_P_weight"]) < 1e-4
    assert rel_err(cp.P.bias.grad, g["grad_P_bias"]) < 1e-4
    assert rel_err(cp.U[2].weight.grad, g["grad_U2_weight"]) < 1e-4
    assert rel_err(yl.grad, g["grad_yl"]) < 1e-4
    with torch.no_grad():
        rows = torch.randn(4, d, device=DEV)
        codes, probs = cp.sample_from_rows(rows, forced_codes=torch.zeros(4, nc, dtype=torch.int64), return_probs=True)
        ref = torch.softmax(cp.P(torch.relu(cp.U[0](cp.layer_norm(rows)))) @ emb[0].weight.t(), 1)
    assert tuple(probs.shape) == (4, nc, nv) and rel_err(probs[:, 0], ref) < 1e-4
