#!/usr/bin/env python
"""Generate the golden fixtures in this directory by running the REAL reference.

Runs only in the build container, where the reference is mounted read-only at /root/reference.
It imports `vidgen` through `oracle/shim` (a stand-in for the un-installed fvcore / termcolor),
loads seeded weights (tests/golden/seeded.py) into the reference's own modules, runs them on
seeded inputs on CPU and stores inputs + outputs as small .npz files.  No reference source,
bytecode or pickled reference object is written anywhere: fixtures are plain arrays.

    python tests/golden/make_golden.py            # regenerate everything
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LVT_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "oracle", "shim"))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import seeded  # noqa: E402

torch.manual_seed(0)
META = {"torch": torch.__version__, "threads": torch.get_num_threads(),
        "mkldnn": bool(torch.backends.mkldnn.is_available())}


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    out["_meta"] = np.frombuffer(json.dumps(META).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


def ref_cfg(path, **over):
    from vidgen.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, path))
    cfg.MODEL.DEVICE = "cpu"
    for k, v in over.items():
        node = cfg
        ks = k.split(".")
        for s in ks[:-1]:
            node = node[s]
        node[ks[-1]] = v
    return cfg


def load_into(module, params):
    missing, unexpected = module.load_state_dict(params, strict=False)
    assert not unexpected, unexpected
    # everything missing must be a non-parameter buffer
    pnames = {n for n, _ in module.named_parameters()}
    assert not (set(missing) & pnames), set(missing) & pnames


def dealias_codebook(cb, state):
    """Give the reference codebook GPU semantics on CPU: running_sum gets its own storage."""
    for i, ve in enumerate(cb.ve):
        ve.embedding.weight.data = state["ve.%d.embedding.weight" % i].clone()
        ve.running_size = state["ve.%d.running_size" % i].clone()
        ve.running_sum = state["ve.%d.running_sum" % i].clone()


def cb_state_of(cb):
    return {k: v.detach().clone() for k, v in cb.state_dict().items()}


# ------------------------------------------------------------------------------------------------
def golden_vqvae():
    from vidgen.modeling.meta_arch.build import build_model
    import vidgen.modeling.meta_arch  # noqa: F401  (registers the meta-archs)
    from vidgen.utils.events import EventStorage

    SEED = 1234
    cfg = ref_cfg("configs/vqvae/PR-DVQVAE2.yaml")
    model = build_model(cfg)
    enc = seeded.seeded_params(seeded.VQVAE_ENCODER_SHAPES, SEED, "enc.")
    dec = seeded.seeded_params(seeded.VQVAE_DECODER_SHAPES, SEED, "dec.")
    load_into(model.encoder, enc)
    load_into(model.generator, dec)

    # G1 encoder ---------------------------------------------------------------------------------
    x = seeded.seeded_input("g1.x", (2, 3, 64, 64), SEED, -1.0, 1.0)
    with torch.no_grad():
        z_e = model.encoder(x.clone())
    save("g1_encoder", seed=SEED, x=x, z_e=z_e)
    zstd = float(z_e.std())

    # G2 vq nearest (standalone function, two codebook regimes) -----------------------------------
    from vidgen.modeling.vq.vq_utils import vq
    rows = z_e.permute(0, 2, 3, 1).contiguous()[..., :64].contiguous()      # (2,16,16,64)
    cb_n = torch.from_numpy(seeded.seeded_array("g2.embedding", (512, 64), SEED)) * zstd
    idx_n = vq(rows, cb_n)
    cb_u = torch.from_numpy(seeded.seeded_input("g2.u", (512, 64), SEED, -1 / 512, 1 / 512).numpy())
    idx_u = vq(rows, cb_u)   # reference's *initial* codebook regime: near-ties everywhere
    save("g2_vq", seed=SEED, rows=rows, cb_normal=cb_n, idx_normal=idx_n, cb_uniform=cb_u, idx_uniform=idx_u)

    # G3 DVQ 'st' one EMA step, de-aliased (GPU semantics) and aliased (reference-on-CPU quirk) ----
    state0 = seeded.seeded_codebook_state(SEED, scale=zstd)
    dealias_codebook(model.codebook, state0)
    with torch.no_grad():
        z_q_st, z_q_bar = model.codebook(z_e, "st")
    st1 = cb_state_of(model.codebook)
    # aliased: fresh module state where running_sum IS the weight storage (as constructed on CPU)
    for i, ve in enumerate(model.codebook.ve):
        ve.embedding.weight.data = state0["ve.%d.embedding.weight" % i].clone()
        ve.running_size = state0["ve.%d.running_size" % i].clone()
        ve.running_sum = ve.embedding.weight.detach()
    with torch.no_grad():
        a_st, a_bar = model.codebook(z_e, "st")
    st1a = cb_state_of(model.codebook)
    with torch.no_grad():
        dealias_codebook(model.codebook, state0)
        idx = model.codebook(z_e)          # mode "" -> (N,4,16,16)
    save("g3_dvq_st", seed=SEED, scale=zstd, z_e=z_e, z_q_st=z_q_st, z_q_bar=z_q_bar, idx=idx,
         aliased_z_q_bar=a_bar,
         **{"new." + k: v for k, v in st1.items()}, **{"aliased." + k: v for k, v in st1a.items()})

    # G4 decoder ---------------------------------------------------------------------------------
    z = seeded.seeded_input("g4.z", (2, 256, 16, 16), SEED, -1.0, 1.0) * zstd
    with torch.no_grad():
        xt = model.generator(z.clone())
    save("g4_decoder", seed=SEED, scale=zstd, z=z, x_tilde=xt)

    # G5 full supervised loss + grads, B=2 frames and B=1 clip of 16 frames ------------------------
    # The clip input is chosen (first of a seeded family) so that EVERY one of its 16 x 1024 code searches has a clear
    # margin between the nearest and the second-nearest code: the fixture then pins losses, gradients and EMA state
    # for any correct fp32 implementation, independent of how sub-margin ties happen to round.
    sys.path.insert(0, ROOT)
    from oracle import lvt_oracle as O
    clip_name = None
    for v in range(400):
        cand = seeded.seeded_input("g5.c%d" % v, (16, 3, 64, 64), SEED)
        with torch.no_grad():
            z = model.encoder(model.normalizer(cand))
        worst = 1.0
        for i in range(4):
            rows = z[:, 64 * i:64 * (i + 1)].permute(0, 2, 3, 1).reshape(-1, 64)
            cb = state0["ve.%d.embedding.weight" % i]
            d0, d1, _ = O.vq_margin_fp64(rows, cb)
            scale = (rows.double() ** 2).sum(-1) + (cb.double() ** 2).sum(-1).max()
            worst = min(worst, float(((d1 - d0) / scale).min()))
        print("  candidate %d: smallest relative margin %.2e" % (v, worst))
        if worst > 8e-6:
            clip_name = "g5.c%d" % v
            print("G5 clip input: %s (smallest relative top-2 margin %.2e)" % (clip_name, worst))
            break
    assert clip_name is not None
    for tag, data in (("frames", [{"image": seeded.seeded_input("g5.f%d" % i, (3, 64, 64), SEED).numpy()}
                                  for i in range(2)]),
                      ("clip", [{"image_sequence": seeded.seeded_input(clip_name, (16, 3, 64, 64), SEED).numpy()}])):
        dealias_codebook(model.codebook, state0)
        model.zero_grad()
        model.train()
        with EventStorage(0):
            losses = model(data, mode="supervised")
        sum(losses.values()).backward()
        g = {n: p.grad.clone() for n, p in list(model.encoder.named_parameters()) +
             [("G." + n, p) for n, p in model.generator.named_parameters()]}
        st = cb_state_of(model.codebook)
        save("g5_vqvae_loss_" + tag, seed=SEED, scale=zstd, input_name=np.array(clip_name if tag == "clip" else "g5.f"),
             loss_reconstruction=losses["loss_reconstruction"], loss_commitment=losses["loss_commitment"],
             grad_enc_first=g["layers.0.weight"], grad_enc_first_bias=g["layers.0.bias"],
             grad_enc_last=g["layers.6.block.3.weight"], grad_enc_mid_rows=g["layers.4.weight"][:8],
             grad_dec_first_rows=g["G.layers.0.weight"][:8], grad_dec_last=g["G.layers.6.weight"],
             grad_dec_last_bias=g["G.layers.6.bias"], grad_dec_ct1_rows=g["G.layers.4.weight"][:4],
             grad_norms=np.array([float(v.norm()) for v in g.values()]),
             grad_names=np.array(list(g.keys())),
             **{"new." + k: v for k, v in st.items() if "ve.0." in k})

    # G22 CODEBOOK.EMA False: the codebook is a trained parameter (vqvae.py:83-84, vq_embedding.py:61-64) -----------
    cfg_ne = ref_cfg("configs/vqvae/PR-DVQVAE2.yaml", **{"MODEL.CODEBOOK.EMA": False})
    model_ne = build_model(cfg_ne)
    load_into(model_ne.encoder, enc)
    load_into(model_ne.generator, dec)
    for i, ve in enumerate(model_ne.codebook.ve):
        ve.embedding.weight.data = state0["ve.%d.embedding.weight" % i].clone()
    model_ne.train()
    model_ne.zero_grad()
    data = [{"image": seeded.seeded_input("g5.f%d" % i, (3, 64, 64), SEED).numpy()} for i in range(2)]
    with EventStorage(0):
        losses = model_ne(data, mode="supervised")
    assert sorted(losses) == ["loss_commitment", "loss_dict", "loss_reconstruction"], sorted(losses)
    sum(losses.values()).backward()
    cbg = {n: p.grad.clone() for n, p in model_ne.codebook.named_parameters()}
    with torch.no_grad():
        idx_ne = model_ne.codebook(model_ne.encoder(model_ne.normalizer(torch.stack([torch.from_numpy(d["image"]) for d in data]))))
    save("g22_vqvae_no_ema", seed=SEED, scale=zstd, loss_reconstruction=losses["loss_reconstruction"],
         loss_commitment=losses["loss_commitment"], loss_dict=losses["loss_dict"], idx=idx_ne,
         grad_enc_first=model_ne.encoder.layers[0].weight.grad, grad_dec_last_bias=model_ne.generator.layers[6].bias.grad,
         generator_param_count=np.array(len(model_ne._generator_parameters())),
         **{"grad." + n: g for n, g in cbg.items()})

    # G6 inference on the five example frames ------------------------------------------------------
    from PIL import Image
    imgs = np.stack([np.asarray(Image.open(os.path.join(REF, "example", "%d.png" % i)).convert("RGB"))
                     for i in range(5)]).transpose(0, 3, 1, 2)              # (5,3,64,64) uint8
    x01 = imgs.astype("float32") / 255.0
    dealias_codebook(model.codebook, state0)
    model.eval()
    with torch.no_grad():
        out = model([{"image_sequence": x01}], mode="inference")[0]
        z_e6 = model.encoder(model.normalizer(torch.from_numpy(x01)))
    save("g6_inference", seed=SEED, scale=zstd, frames_u8=imgs, latent=out["latent"],
         reconstruction=out["reconstruction"], z_e=z_e6)


# ------------------------------------------------------------------------------------------------
def golden_vt():
    from vidgen.modeling.meta_arch.build import build_model
    import vidgen.modeling.meta_arch  # noqa: F401
    from vidgen.modeling.autoregressive import vt_utils
    from vidgen.modeling.autoregressive.vt_attention import PositionalEncoding
    from vidgen.data.dataset_mapper import DatasetMapper
    from vidgen.utils.events import EventStorage
    import random

    SEED = 4321
    # G7 subscale helpers: DSFVT geometry (16,1,1)/(7,1,1) for all a, and a (4,2,2)/(3,3,3) one ------
    vid = seeded.seeded_codes("g7.video", (1, 4, 16, 16, 16), SEED)
    g7 = {"video": vid}
    for a in range(16):
        vm = vt_utils.visible_abc_mask(a, 0, 0, 16, 1, 1, 16, 16, 16, dtype=torch.bool)
        ctx = vt_utils.ss_shift(vid.masked_fill(~vm, -1), a, 0, 0, 16, 1, 1, 16, 16, 16, 7, 1, 1, pad_value=-1)
        g7["dsfvt_ctx_%d" % a] = ctx
    vid2 = seeded.seeded_codes("g7.video2", (1, 2, 8, 8, 8), SEED)
    g7["video2"] = vid2
    for (a, b, c) in ((0, 0, 0), (1, 0, 1), (3, 1, 1), (2, 1, 0)):
        sm = vt_utils.slice_mask(a, b, c, 4, 2, 2, 8, 8, 8, dtype=torch.bool)
        vm = vt_utils.visible_abc_mask(a, b, c, 4, 2, 2, 8, 8, 8, dtype=torch.bool)
        ctx = vt_utils.ss_shift(vid2.masked_fill(~vm, -1), a, b, c, 4, 2, 2, 8, 8, 8, 3, 3, 3, pad_value=-1)
        g7["g422_smask_%d%d%d" % (a, b, c)] = sm
        g7["g422_vmask_%d%d%d" % (a, b, c)] = vm
        g7["g422_ctx_%d%d%d" % (a, b, c)] = ctx
    save("g7_subscale", seed=SEED, **g7)

    # G8 DatasetMapper prepare_slices for forced (a,b,c) ---------------------------------------------
    cfg = ref_cfg("configs/vt/DSFVT.yaml")
    mapper = DatasetMapper(cfg, True)
    codes = seeded.seeded_codes("g8.codes", (16, 4, 16, 16), SEED).numpy()
    g8 = {"codes": codes}
    real_randint = random.randint
    for a in (1, 2, 5, 9, 15):
        seq = iter([0, a, 0, 0])  # start_end() draws first (dataset_mapper.py:44), then a, b, c
        random.randint = lambda lo, hi: next(seq)
        try:
            d = mapper({"image_sequence": codes.copy()})
        finally:
            random.randint = real_randint
        for k in ("context", "slice", "slice_idx", "ignore_mask"):
            g8["a%d_%s" % (a, k)] = d[k]
    save("g8_mapper", seed=SEED, **g8)

    # model with seeded weights ------------------------------------------------------------------
    model = build_model(cfg)
    params = seeded.seeded_params(seeded.dsfvt_shapes(), SEED)
    load_into(model.model, params)
    vt = model.model

    # G9 masked conv, positional table, get_B -----------------------------------------------------
    xe = seeded.seeded_input("g9.x", (2, 128, 1, 16, 16), SEED, -1.0, 1.0)
    with torch.no_grad():
        yc = vt.decoder.conv(xe)
        wmasked = vt.decoder.conv.conv.weight.detach().clone()
        pe = PositionalEncoding(512)(torch.zeros(1, 512, 1, 16, 16))
        pe3 = PositionalEncoding(48)(torch.zeros(1, 48, 3, 4, 5))
        B = vt.decoder.block_local_attention[0].get_B()
    save("g9_pieces", seed=SEED, x=xe, masked_conv_out=yc, masked_taps=wmasked[:4, :4],
         pos_table=pe[0], pos_table_48_345=pe3[0], B_dec0_head3=B[3, 0], B_dec0_corner=B[:, 0, :4, :4])

    # G10 one BlockLocalAttention layer, masked and unmasked, fwd + grads ---------------------------
    for tag, layer in (("masked", vt.decoder.block_local_attention[0]),
                       ("unmasked", vt.encoder.block_local_attention[0])):
        x = seeded.seeded_input("g10.x." + tag, (2, 512, 1, 16, 16), SEED, -1.0, 1.0).requires_grad_(True)
        gy = seeded.seeded_input("g10.gy." + tag, (2, 512, 1, 16, 16), SEED, -1.0, 1.0)
        vt.zero_grad()
        y = layer(x)
        y.backward(gy)
        save("g10_bla_" + tag, seed=SEED, x=x, gy=gy, y=y, grad_x=x.grad,
             grad_w_q_h0=layer.mha.w_q.grad[0, :, :16], grad_w_v_h7=layer.mha.w_v.grad[7, :16],
             grad_proj_rows=layer.mha.proj.weight.grad[:8], grad_dh_bank=layer.dh_bank.grad,
             grad_dw_bank=layer.dw_bank.grad, grad_dt_bank=layer.dt_bank.grad,
             grad_ln_w=layer.mha.layer_norm.weight.grad, grad_ln_b=layer.mha.layer_norm.bias.grad,
             grad_ffn1_rows=layer.ffn[1].weight.grad[:8], grad_ffn3_b=layer.ffn[3].bias.grad,
             grad_ffn0_w=layer.ffn[0].weight.grad)

    # G11 channel predictor logits ----------------------------------------------------------------
    sl = seeded.seeded_codes("g11.slice", (2, 4, 1, 16, 16), SEED)
    yl = seeded.seeded_input("g11.yl", (2, 512, 1, 16, 16), SEED, -2.0, 2.0)
    with torch.no_grad():
        pred = vt.ch_predictor(sl, yl, mode="logits")
    save("g11_chpred", seed=SEED, slice=sl, yl=yl, **{"logits_%d" % k: pred[k][:, :, 0, ::5, ::3] for k in range(4)},
         logits_3_full_b0=pred[3][0])

    # G12 full DSFVT supervised loss + grads at b=2 -------------------------------------------------
    codes2 = [seeded.seeded_codes("g12.codes%d" % i, (16, 4, 16, 16), SEED).numpy() for i in range(2)]
    data = []
    for i, a in enumerate((3, 11)):
        seq = iter([0, a, 0, 0])  # start_end() draws first (dataset_mapper.py:44), then a, b, c
        random.randint = lambda lo, hi: next(seq)
        try:
            data.append(mapper({"image_sequence": codes2[i].copy()}))
        finally:
            random.randint = real_randint
    model.train()
    vt.zero_grad()
    with EventStorage(0):
        losses = model(data, mode="supervised")
    losses["loss_cross_entropy"].backward()
    with torch.no_grad():
        ctx = torch.stack([d["context"] for d in data])
        slc = torch.stack([d["slice"] for d in data])
        sidx = torch.stack([d["slice_idx"] for d in data])
        zl = vt.encoder(ctx, sidx)
        yl2 = vt.decoder(slc, zl)
        pred = vt.ch_predictor(slc, yl2, mode="logits")
    names = [n for n, _ in vt.named_parameters()]
    norms = np.array([float(p.grad.norm()) for _, p in vt.named_parameters()])
    gd = dict(vt.named_parameters())
    save("g12_dsfvt_loss", seed=SEED, codes=np.stack(codes2), a=np.array([3, 11]),
         loss=losses["loss_cross_entropy"], zl_slice=zl[:, ::16, 0, ::4, ::4], yl_slice=yl2[:, ::16, 0, ::4, ::4],
         logits0_slice=pred[0][:, ::8, 0, ::4, ::4], logits3_slice=pred[3][:, ::8, 0, ::4, ::4],
         grad_names=np.array(names), grad_norms=norms,
         grad_enc_conv_rows=gd["encoder.conv.weight"].grad[:2, :, :, 0, 0],
         grad_slice_emb=gd["encoder.slice_embedding.weight"].grad,
         grad_ch_emb0_rows=gd["decoder.ch_embedder.0.weight"].grad[:16],
         grad_dec_conv_rows=gd["decoder.conv.conv.weight"].grad[:2],
         grad_U3_rows=gd["ch_predictor.U.3.weight"].grad[:2],
         grad_P0_bias=gd["ch_predictor.P.0.bias"].grad,
         grad_dec7_dh=gd["decoder.block_local_attention.7.dh_bank"].grad,
         grad_enc0_wq_h0=gd["encoder.block_local_attention.0.mha.w_q"].grad[0, :8])

    # G13 sample_pixel probabilities for three pixels ----------------------------------------------
    model.eval()
    g13 = {}
    with torch.no_grad():
        b0 = 0
        ctx1, sl1, si1 = ctx[b0:b0 + 1], slc[b0:b0 + 1].clone(), sidx[b0:b0 + 1]
        zl1 = vt.encoder(ctx1, si1)
        for (hi, wi) in ((0, 0), (7, 9), (15, 15)):
            yl1 = vt.decoder(sl1, zl1)
            y = vt.ch_predictor.layer_norm(yl1[:, :, 0, hi, wi])
            # teacher-forced probabilities: feed the slice's own codes as the "previous draws"
            oh = torch.nn.functional.one_hot(sl1[:, :, 0, hi, wi], 512).float().view(1, -1)
            pr = []
            for k in range(4):
                inp = y if k == 0 else torch.cat((y, oh[:, :k * 512]), 1)
                o = vt.ch_predictor.P[k](torch.relu(vt.ch_predictor.U[k](inp)))
                pr.append(torch.softmax(o / 1.0, 1))
            g13["probs_%d_%d" % (hi, wi)] = torch.stack(pr, 1)[0]
    save("g13_sample_probs", seed=SEED, a=3, **g13)

    # G14 logits for an entire video (BitsEvaluator input) -----------------------------------------
    cfg2 = ref_cfg("configs/vt/DSFVT.yaml")
    cfg2.TEST.EVALUATORS = "BitsEvaluator"
    model.cfg = cfg2
    with torch.no_grad():
        out = model([{"image_sequence": torch.from_numpy(codes2[0])}], mode="inference")[0]
    lg = out["logits"]                                                  # (4,512,16,16,16)
    tgt = torch.from_numpy(codes2[0]).transpose(0, 1)                   # (4,16,16,16)
    nll = torch.nn.functional.cross_entropy(lg.permute(1, 0, 2, 3, 4)[None], tgt[None], reduction="none")[0]
    save("g14_video_logits", seed=SEED, logits_slice=lg[:, ::64, ::3, ::5, ::5], nll=nll,
         ignore_mask=out["ignore_mask"])


# ------------------------------------------------------------------------------------------------
def golden_variants():
    """G15-G18: the other shipped shapes -- DSSVT (spatial subscaling, block-split attention at test length),
    DSTSVT (spatio-temporal subscaling), class-conditional DSFVT (CLASS_NUM > 0), K-DVQVAE (4 residual blocks)."""
    from vidgen.modeling.meta_arch.build import build_model
    import vidgen.modeling.meta_arch  # noqa: F401
    from vidgen.data.dataset_mapper import DatasetMapper
    from vidgen.utils.events import EventStorage
    import random

    SEED = 777
    real_randint = random.randint

    def vt_case(tag, cfg_path, block, kernel, n_slices, frames, abcs, class_num=0, classes=None, eval_frames=0):
        over = {"MODEL.AUTOREGRESSIVE.VT.CLASS_NUM": class_num} if class_num else {}
        cfg = ref_cfg(cfg_path, **over)
        model = build_model(cfg)
        vt = model.model
        params = seeded.seeded_params(seeded.dsfvt_shapes(block=block, kernel=kernel, n_slices=n_slices,
                                                          class_num=class_num), SEED)
        load_into(vt, params)
        mapper = DatasetMapper(cfg, True)
        codes = [seeded.seeded_codes("%s.codes%d" % (tag, i), (frames, 4, 16, 16), SEED).numpy() for i in range(len(abcs))]
        data = []
        for i, (a, b, c) in enumerate(abcs):
            seq = iter([0, a, b, c])          # start_end() draws first (dataset_mapper.py:44), then a, b, c
            random.randint = lambda lo, hi: next(seq)
            try:
                d = {"image_sequence": codes[i].copy()}
                if classes is not None:
                    d["class"] = int(classes[i])
                data.append(mapper(d))
            finally:
                random.randint = real_randint
        model.train()
        vt.zero_grad()
        with EventStorage(0):
            losses = model(data, mode="supervised")
        losses["loss_cross_entropy"].backward()
        with torch.no_grad():
            ctx = torch.stack([d["context"] for d in data])
            slc = torch.stack([d["slice"] for d in data])
            sidx = torch.stack([d["slice_idx"] for d in data])
            cls = torch.stack([d["class"] for d in data]) if classes is not None else None
            zl = vt.encoder(ctx, sidx, class_idx=cls)
            pred = vt.ch_predictor(slc, vt.decoder(slc, zl), mode="logits")
        names = [n for n, _ in vt.named_parameters()]
        gd = dict(vt.named_parameters())
        out = dict(seed=SEED, codes=np.stack(codes), abc=np.array(abcs), loss=losses["loss_cross_entropy"],
                   context=ctx, slice_idx=sidx, zl_slice=zl[:, ::16, :, ::3, ::3],
                   logits0_slice=pred[0][:, ::8, :, ::3, ::3], logits3_slice=pred[3][:, ::8, :, ::3, ::3],
                   grad_names=np.array(names), grad_norms=np.array([float(p.grad.norm()) for _, p in vt.named_parameters()]),
                   grad_enc_conv_rows=gd["encoder.conv.weight"].grad[:2],
                   grad_slice_emb=gd["encoder.slice_embedding.weight"].grad,
                   grad_dec3_dt=gd["decoder.block_local_attention.3.dt_bank"].grad,
                   grad_enc_proj_rows=gd["encoder.linear_projector.weight"].grad[:4, :, 0, 0, 0])
        if classes is not None:
            out["classes"] = np.array(classes)
            out["grad_class_emb"] = gd["encoder.class_embedding.weight"].grad
        if eval_frames:
            cfg.TEST.EVALUATORS = "BitsEvaluator"
            model.eval()
            vid = seeded.seeded_codes(tag + ".eval", (eval_frames, 4, 16, 16), SEED)
            with torch.no_grad():
                o = model([{"image_sequence": vid}], mode="inference")[0]
            lg = o["logits"]
            nll = torch.nn.functional.cross_entropy(lg.permute(1, 0, 2, 3, 4)[None], vid.transpose(0, 1)[None],
                                                    reduction="none")[0]
            out.update(eval_video=vid, eval_logits_slice=lg[:, ::64, ::3, ::5, ::5], eval_nll=nll,
                       eval_ignore_mask=o["ignore_mask"])
        save(tag, **out)

    # G15 DSSVT: stride (1,2,2), kernel (1,3,3), 4-frame training clips -> slices (4,8,8); at the 16-frame test
    # length the slice is (16,8,8) and every attention layer runs block-split over four (4,8,8) blocks
    vt_case("g15_dssvt", "configs/vt/DSSVT.yaml", (4, 8, 8), (1, 3, 3), 4, 4, [(0, 1, 0), (0, 1, 1)], eval_frames=16)
    # G16 DSTSVT: stride (4,2,2), kernel (5,3,3), 16-frame clips -> slices (4,8,8)
    vt_case("g16_dstsvt", "configs/vt/DSTSVT.yaml", (4, 8, 8), (5, 3, 3), 16, 16, [(2, 1, 0), (3, 0, 1)])
    # G17 class-conditional DSFVT (CLASS_NUM 10)
    vt_case("g17_dsfvt_class", "configs/vt/DSFVT.yaml", (1, 16, 16), (7, 1, 1), 16, 16, [(5, 0, 0), (12, 0, 0)],
            class_num=10, classes=[3, 7])

    # G18 K-DVQVAE: 4 residual blocks in encoder and decoder, frame mode, B = 2 frames
    cfg = ref_cfg("configs/vqvae/K-DVQVAE.yaml")
    model = build_model(cfg)
    es, ds = seeded.vqvae_shapes(4)
    load_into(model.encoder, seeded.seeded_params(es, SEED, "enc."))
    load_into(model.generator, seeded.seeded_params(ds, SEED, "dec."))
    state = seeded.seeded_codebook_state(SEED, scale=0.6)
    dealias_codebook(model.codebook, state)
    x = seeded.seeded_input("g18.x", (2, 3, 64, 64), SEED)
    model.train()
    with EventStorage(0):
        losses = model([{"image": x[i].numpy()} for i in range(2)], mode="supervised")
    sum(losses.values()).backward()
    ge, gg = dict(model.encoder.named_parameters()), dict(model.generator.named_parameters())
    with torch.no_grad():
        z_e = model.encoder(model.normalizer(x))
    save("g18_kdvqvae", seed=SEED, x=x, z_e_slice=z_e[:, ::8, ::2, ::2],
         **{k: v for k, v in losses.items()},
         enc_grad_norms=np.array([float(p.grad.norm()) for p in ge.values()]), enc_grad_names=np.array(list(ge)),
         dec_grad_norms=np.array([float(p.grad.norm()) for p in gg.values()]), dec_grad_names=np.array(list(gg)),
         grad_enc_l8_b3=ge["layers.8.block.3.weight"].grad[:8, :, 0, 0], grad_dec_l8_rows=gg["layers.8.weight"].grad[:2],
         **{"new_" + k: v for k, v in cb_state_of(model.codebook).items() if "running_size" in k})


def tensor_pins(state_dict):
    """Per-tensor checksums (float64 sum, sum of |.|, first and last element): what a same-seed construction of the
    product's build_model has to reproduce exactly."""
    names, rows = [], []
    for k, v in state_dict.items():
        if not v.dtype.is_floating_point:
            continue
        d = v.detach().double().reshape(-1)
        names.append(k)
        rows.append([float(d.sum()), float(d.abs().sum()), float(d[0]), float(d[-1]), float(d.numel())])
    return np.array(names), np.array(rows, dtype=np.float64)


LATENT_TREE = {                      # relative leaf directory -> file names (the test rebuilds the same tree)
    "video_0": ["%d.npy" % i for i in range(12)],
    "video_1": ["0.npy", "1.npy", "2.npy", "10.npy", "9.npy"],
    "clsA/video_7": ["3.npy", "1.npy", "2.npy"],
    "clsA/video_8": ["0.npy"],
    "mixed": ["0.npy", "notes.txt"],
    "emptyleaf": [],
}


def golden_pins():
    """G19: weight initialisation (A21) -- checksums of every tensor of a same-seed `build_model`; G20: the loader
    records / `latent_video_paths.npy` cache (f2) on a small directory tree; the mapper fed with such a record."""
    import random
    import tempfile
    from vidgen.modeling.meta_arch.build import build_model
    import vidgen.modeling.meta_arch  # noqa: F401
    from vidgen.data.datasets.latents import get_latent_video_paths
    from vidgen.data.dataset_mapper import DatasetMapper

    out = {}
    for tag, path, seed in (("prdvqvae2", "configs/vqvae/PR-DVQVAE2.yaml", 29871897),
                            ("kdvqvae", "configs/vqvae/K-DVQVAE.yaml", 11),
                            ("dsfvt", "configs/vt/DSFVT.yaml", 29871897),
                            ("dssvt", "configs/vt/DSSVT.yaml", 5)):
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
        model = build_model(ref_cfg(path))
        parts = {"encoder": model.encoder, "generator": model.generator, "codebook": model.codebook} \
            if hasattr(model, "codebook") else {"model": model.model}
        for part, mod in parts.items():
            names, rows = tensor_pins(mod.state_dict())
            out["%s.%s.names" % (tag, part)] = names
            out["%s.%s.pins" % (tag, part)] = rows
        out[tag + ".seed"] = seed
    save("g19_init_pins", **out)

    with tempfile.TemporaryDirectory() as root:
        rng = np.random.RandomState(3)
        for leaf, files in LATENT_TREE.items():
            os.makedirs(os.path.join(root, leaf), exist_ok=True)
            for f in files:
                full = os.path.join(root, leaf, f)
                if f.endswith(".npy"):
                    np.save(full, rng.randint(0, 512, (4, 16, 16)).astype(np.int64))
                else:
                    open(full, "w").write("x")
        recs = get_latent_video_paths(root, use_cache=True)
        assert os.path.exists(os.path.join(root, "latent_video_paths.npy"))
        again = get_latent_video_paths(root, use_cache=True)
        assert again == recs
        rel = lambda p_: os.path.relpath(p_, root)            # noqa: E731
        g20 = {"video_path": np.array([rel(r["video_path"]) for r in recs]),
               "latent_paths": np.array(["|".join(rel(q) for q in r["latent_paths"]) for r in recs]),
               "video_idx": np.array([r["video_idx"] for r in recs]),
               "record_keys": np.array(sorted(recs[0]))}
        # the mapper on a loader record: window + slice drawn with python `random` seeded to 77
        cfg = ref_cfg("configs/vt/DSFVT.yaml")
        cfg.INPUT.N_FRAMES_PER_VIDEO_TRAIN = 8
        cfg.MODEL.AUTOREGRESSIVE.VT.STRIDE = (8, 1, 1)
        cfg.MODEL.AUTOREGRESSIVE.VT.N_PRIME = 2
        mapper = DatasetMapper(cfg, True)
        rec0 = [r for r in recs if rel(r["video_path"]) == "video_0"][0]
        random.seed(77)
        d = mapper(rec0)
        g20.update(mapped_context=d["context"], mapped_slice=d["slice"], mapped_slice_idx=d["slice_idx"],
                   mapped_ignore=d["ignore_mask"], mapped_keys=np.array(sorted(k for k in d)),
                   video_0=np.stack([np.load(q) for q in rec0["latent_paths"]]))
    save("g20_latent_paths", **g20)


def golden_share_p():
    """G21: ChannelPredictor with ONE shared output layer P (SHARE_P = True, the reference's config default:
    vidgen/config/defaults.py:50, videotransformer.py:121-123,150-151): logits of all channels and the gradients of the
    shared P (the sum over the channels), of U_2 and of the input, at reduced width (d = 128, nv = 64, nc = 3)."""
    from vidgen.modeling.autoregressive.videotransformer import ChannelPredictor
    SEED = 2121
    d, nc, nv, de = 128, 3, 64, 32
    cp = ChannelPredictor(d, nc, nv, de, share_p=True, share_embeddings=False)
    shapes = {"layer_norm.weight": (d,), "layer_norm.bias": (d,), "P.weight": (nv, d), "P.bias": (nv,)}
    for k in range(nc):
        shapes["U.%d.weight" % k] = (d, d + k * nv)
        shapes["U.%d.bias" % k] = (d,)
    assert set(shapes) == set(cp.state_dict().keys())
    params = seeded.seeded_params(shapes, SEED, "g21.")
    load_into(cp, params)
    sl = seeded.seeded_codes("g21.slice", (2, nc, 2, 8, 8), SEED, nv=nv)
    yl = seeded.seeded_input("g21.yl", (2, d, 2, 8, 8), SEED, -2.0, 2.0).requires_grad_(True)
    gys = [seeded.seeded_input("g21.gy%d" % k, (2, nv, 2, 8, 8), SEED, -1.0, 1.0) for k in range(nc)]
    pred = cp(sl, yl, mode="logits")
    sum((o * g).sum() for o, g in zip(pred, gys)).backward()
    save("g21_share_p", seed=SEED, dims=np.array([d, nc, nv, de]), slice=sl, yl=yl,
         **{"logits_%d" % k: pred[k] for k in range(nc)}, **{"gy_%d" % k: gys[k] for k in range(nc)},
         grad_P_weight=cp.P.weight.grad, grad_P_bias=cp.P.bias.grad, grad_U2_weight=cp.U[2].weight.grad,
         grad_U0_bias=cp.U[0].bias.grad, grad_yl=yl.grad, grad_ln_w=cp.layer_norm.weight.grad)


def golden_single_codebook():
    """G23: CODEBOOK.NUM == 1, the default of the config tree (config/defaults.py:79) and what configs/vqvae/Base-VQVAE.yaml
    builds on its own: ONE 256-d codebook, `VQEmbedding` used directly (meta_arch/vqvae.py:25-27, vq_embedding.py:9-66)."""
    from vidgen.modeling.meta_arch.build import build_model
    import vidgen.modeling.meta_arch  # noqa: F401
    from vidgen.utils.events import EventStorage
    SEED = 1234
    # (Base-VQVAE.yaml alone is a 1-channel model; PR-DVQVAE2 = Base + the 3-channel image ends and NUM 4: take it with NUM 1)
    assert ref_cfg("configs/vqvae/Base-VQVAE.yaml").MODEL.CODEBOOK.NUM == 1
    cfg = ref_cfg("configs/vqvae/PR-DVQVAE2.yaml", **{"MODEL.CODEBOOK.NUM": 1})
    model = build_model(cfg)
    assert type(model.codebook).__name__ == "VQEmbedding"
    load_into(model.encoder, seeded.seeded_params(seeded.VQVAE_ENCODER_SHAPES, SEED, "enc."))
    load_into(model.generator, seeded.seeded_params(seeded.VQVAE_DECODER_SHAPES, SEED, "dec."))
    x = seeded.seeded_input("g1.x", (2, 3, 64, 64), SEED, -1.0, 1.0)
    with torch.no_grad():
        zstd = float(model.encoder(x.clone()).std())
    st = seeded.seeded_codebook_state(SEED, num=1, K=512, D=256, scale=zstd)
    state0 = {k[len("ve.0."):]: v for k, v in st.items()}

    def dealias():
        model.codebook.embedding.weight.data = state0["embedding.weight"].clone()
        model.codebook.running_size = state0["running_size"].clone()
        model.codebook.running_sum = state0["running_sum"].clone()
    data = [{"image": seeded.seeded_input("g5.f%d" % i, (3, 64, 64), SEED).numpy()} for i in range(2)]
    xin = model.normalizer(torch.stack([torch.from_numpy(d["image"]) for d in data]))
    dealias()
    with torch.no_grad():
        z_e = model.encoder(xin.clone())
        idx = model.codebook(z_e)                          # mode "": (N, 16, 16)
        lat = model.encode(xin.clone())
        dec = model.decode(lat)
    assert torch.equal(idx, lat)
    dealias()
    model.train()
    model.zero_grad()
    with EventStorage(0):
        losses = model(data, mode="supervised")
    assert sorted(losses) == ["loss_commitment", "loss_reconstruction"]
    sum(losses.values()).backward()
    new = cb_state_of(model.codebook)
    save("g23_single_codebook", seed=SEED, scale=zstd, state_keys=np.array(sorted(new.keys())), z_e=z_e, idx=idx,
         decode_slice=dec[:, :, ::4, ::4],
         loss_reconstruction=losses["loss_reconstruction"], loss_commitment=losses["loss_commitment"],
         grad_enc_first=model.encoder.layers[0].weight.grad, grad_enc_first_bias=model.encoder.layers[0].bias.grad,
         grad_dec_last=model.generator.layers[6].weight.grad, grad_dec_last_bias=model.generator.layers[6].bias.grad,
         **{"new." + k: v for k, v in new.items()})


def golden_share_embeddings():
    """G24: ChannelPredictor with SHARE_EMBEDDINGS (videotransformer.py:124-125,152-154): ONE layer P: d -> de whose output meets
    the decoder's channel embedding table E_k as the output matrix (logits_k = P(relu(u_k)) E_k^T).  Logits, and the gradients of P,
    of the tied tables, of U_2 and of the input, at reduced width (d = 128, nv = 64, nc = 3, de = 32)."""
    from vidgen.modeling.autoregressive.videotransformer import ChannelPredictor
    SEED = 2424
    d, nc, nv, de = 128, 3, 64, 32
    cp = ChannelPredictor(d, nc, nv, de, share_p=False, share_embeddings=True)
    shapes = {"layer_norm.weight": (d,), "layer_norm.bias": (d,), "P.weight": (de, d), "P.bias": (de,)}
    for k in range(nc):
        shapes["U.%d.weight" % k] = (d, d + k * nv)
        shapes["U.%d.bias" % k] = (d,)
    assert set(shapes) == set(cp.state_dict().keys())
    load_into(cp, seeded.seeded_params(shapes, SEED, "g24."))
    emb = torch.nn.ModuleList([torch.nn.Embedding(nv, de) for _ in range(nc)])
    load_into(emb, seeded.seeded_params({"%d.weight" % k: (nv, de) for k in range(nc)}, SEED, "g24.emb."))
    sl = seeded.seeded_codes("g24.slice", (2, nc, 2, 8, 8), SEED, nv=nv)
    yl = seeded.seeded_input("g24.yl", (2, d, 2, 8, 8), SEED, -2.0, 2.0).requires_grad_(True)
    gys = [seeded.seeded_input("g24.gy%d" % k, (2, nv, 2, 8, 8), SEED, -1.0, 1.0) for k in range(nc)]
    pred = cp(sl, yl, mode="logits", ch_embedder=emb)
    sum((o * g).sum() for o, g in zip(pred, gys)).backward()
    save("g24_share_embeddings", seed=SEED, dims=np.array([d, nc, nv, de]), slice=sl, yl=yl,
         **{"logits_%d" % k: pred[k] for k in range(nc)}, **{"gy_%d" % k: gys[k] for k in range(nc)},
         grad_P_weight=cp.P.weight.grad, grad_P_bias=cp.P.bias.grad, grad_U2_weight=cp.U[2].weight.grad,
         grad_yl=yl.grad, **{"grad_emb_%d" % k: emb[k].weight.grad for k in range(nc)})


if __name__ == "__main__":
    which = sys.argv[1:] or ["vqvae", "vt", "variants", "pins", "share_p", "single", "share_emb"]
    if "share_p" in which:
        golden_share_p()
    if "single" in which:
        golden_single_codebook()
    if "share_emb" in which:
        golden_share_embeddings()
    if "pins" in which:
        golden_pins()
    if "vqvae" in which:
        golden_vqvae()
    if "vt" in which:
        golden_vt()
    if "variants" in which:
        golden_variants()
