"""GPU parity of the plane-fed GEMM path of the f16x2 arithmetic (csrc/gemm_p2.hip, round 6).

A P2 image holds exactly what lvt_gemm_f32 stages in LDS in LVT_MATH_F16X2 mode, the kernel issues the same matrix instructions
in the same order into the same accumulators, so the bar is BIT-IDENTITY with lvt_gemm_f32 on the fp32 matrices the images were
made from (torch.equal) -- which in turn is held to 2e-5 of torch fp32 by tests/test_gpu_engine.py.  The images themselves are
checked byte for byte against a torch restatement of the split (hi = RN16(x s), lo = RN16(2^11 (x s - hi)))."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def f16x2_mode():
    from lvt_amd.hip import binding as L
    before = L.get_math_mode()
    L.set_math_mode("f16x2")
    yield
    L.set_math_mode(before)


def _dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(_dev())


def _scale_of(amax):
    """lvt_f16_scale: the power of two s with amax * s in [2^14, 2^15)."""
    bits = int(torch.tensor([float(amax)], dtype=torch.float32).view(torch.int32))
    eb = (bits >> 23) & 0xff
    se = min(max(268 - eb, 2), 252)
    return float(torch.tensor([se << 23], dtype=torch.int32).view(torch.float32))


def _image_ref(x, amax):
    """(rows, K) fp32 -> (rows, K / 32, 2, 32) fp16: the P2 image, restated with torch ops."""
    s = _scale_of(amax)
    xs = x.double() * s                                   # exact: a power of two
    hi = xs.float().to(torch.float16)
    lo = ((xs - hi.double()) * 2048.0).float().to(torch.float16)
    r, k = x.shape
    return torch.stack([hi.view(r, k // 32, 32), lo.view(r, k // 32, 32)], dim=2)


def _image_view(img):
    r, k = img.shape
    return img.view(torch.float16).view(r, k // 32, 2, 32)


def _pack(x, transpose=False):
    from lvt_amd.hip import gemm as G, binding as L
    rows, k = (x.shape[1], x.shape[0]) if transpose else x.shape
    dst = torch.empty(rows, k, dtype=torch.float32, device=x.device)
    amax = L.amax_of(x)
    G.p2_pack([(x, transpose, dst, amax)])
    return G.P2Image(dst, amax)


@pytest.mark.parametrize("rows,K,transpose", [(64, 64, False), (100, 96, False), (512, 512, False), (128, 512, True), (96, 128, True),
                                              (3072, 512, False)])
def test_p2_pack_bytes(rows, K, transpose):
    x = _rand(K, rows, seed=3) if transpose else _rand(rows, K, seed=3)
    x[0, 0] = 3.7                                     # the scale comes from the max, not from the range of rand
    img = _pack(x, transpose)
    ref = _image_ref(x.t().contiguous() if transpose else x, float(x.abs().max()))
    assert torch.equal(_image_view(img.data), ref)


def test_p2_pack_ladder_and_zero():
    """rows from 1 down to 2^-30 of the max, an all-zero row, denormal-range residuals: the image restates the split exactly."""
    x = _rand(64, 128, seed=5)
    for r in range(64):
        x[r] *= 2.0 ** (-r / 2)
    x[7] = 0
    img = _pack(x)
    assert torch.equal(_image_view(img.data), _image_ref(x, float(x.abs().max())))


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (256, 128, 64), (1000, 384, 96), (4096, 512, 512), (16384, 512, 512), (300, 132, 512)])
@pytest.mark.parametrize("a_image", [False, True])
def test_gemm_p2_equals_engine(M, N, K, a_image):
    from lvt_amd.hip import gemm as G, binding as L
    a, w = _rand(M, K), _rand(N, K, seed=1, scale=0.3)
    b, r = _rand(N, seed=2), _rand(M, N, seed=3)
    ref = torch.empty(M, N, device=_dev())
    G.gemm(a, w, ref, M, N, K, ta=0, tb=0, flags=L.EPI_BIAS | L.EPI_RESIDUAL | L.EPI_RELU, bias=b, res=r)
    out = torch.empty(M, N, device=_dev())
    A = _pack(a) if a_image else a
    G.gemm_p2(A, _pack(w), out, M, N, K, flags=L.EPI_BIAS | L.EPI_RESIDUAL | L.EPI_RELU, bias=b, res=r)
    assert torch.equal(out, ref)
    assert float(L.amax_of(out)) == float(out.abs().max())


def test_gemm_p2_mask_and_plain():
    from lvt_amd.hip import gemm as G, binding as L
    M, N, K = 2048, 512, 512
    a, w, mk = _rand(M, K), _rand(N, K, seed=1), _rand(M, N, seed=4)
    for flags, kw in ((L.EPI_MASK, dict(mask=mk)), (0, {})):
        ref, out = torch.empty(M, N, device=_dev()), torch.empty(M, N, device=_dev())
        G.gemm(a, w, ref, M, N, K, flags=flags, **kw)
        G.gemm_p2(a, _pack(w), out, M, N, K, flags=flags, **kw)
        assert torch.equal(out, ref)


def test_gemm_p2_transposed_weight_image_is_the_nn_form():
    """dx = dy W (the engine's tb = 1 form on W (n_out, k_in)) == gemm_p2 on the image of W^T."""
    from lvt_amd.hip import gemm as G
    M, n_out, k_in = 4096, 512, 1024
    dy, w = _rand(M, n_out), _rand(n_out, k_in, seed=1, scale=0.2)
    ref, out = torch.empty(M, k_in, device=_dev()), torch.empty(M, k_in, device=_dev())
    G.gemm(dy, w, ref, M, k_in, n_out, ta=0, tb=1, ldb=k_in)
    G.gemm_p2(dy, _pack(w, transpose=True), out, M, k_in, n_out)
    assert torch.equal(out, ref)


def test_gemm_p2_qkv_forms():
    """The two q/k/v products of an attention layer on images of the packed (3, na, d, da) weights: forward = 3 x na batches of
    (M x da x d) into C (3, M, hd); data gradient = one (M x d x 3 hd) product whose A walks the (3, M, hd) gradient with a
    two-level k (vt_attention.py:120-124 and its backward)."""
    from lvt_amd.hip import gemm as G, binding as L
    M, d, na, da = 1024, 512, 8, 128
    hd = na * da
    x, w = _rand(M, d), _rand(3, na, d, da, seed=1, scale=0.1)
    amax = L.amax_of(w)
    # forward
    ref = torch.empty(3, M, hd, device=_dev())
    G.gemm(x, w, ref, M, da, d, ta=0, tb=1, lda=d, ldb=da, ldc=hd, batch_outer=3, batch_inner=na, sB=(na * d * da, d * da), sC=(M * hd, da))
    wf = torch.empty(3 * na * da, d, device=_dev())           # image rows (p, h, j), k = d
    G.p2_pack([(w[p_, h_], True, wf[(p_ * na + h_) * da:(p_ * na + h_ + 1) * da], amax) for p_ in range(3) for h_ in range(na)])
    wf2 = torch.empty_like(wf)                                # the same image from ONE batched entry
    G.p2_pack([(w.view(3 * na * d, da)[:d], True, wf2[:da], amax, 3 * na, d * da, da * d)])
    assert torch.equal(wf, wf2)
    out = torch.empty(3, M, hd, device=_dev())
    G.gemm_p2(x, G.P2Image(wf, amax), out, M, da, d, lda=d, ldb=d, ldc=hd, batch_outer=3, batch_inner=na,
              sB=(na * da * d, da * d), sC=(M * hd, da))
    assert torch.equal(out, ref)
    # data gradient
    g = _rand(3, M, hd, seed=2)
    ref2 = torch.empty(M, d, device=_dev())
    G.gemm(g, w, ref2, M, d, 3 * hd, ta=0, tb=0, lda=hd, a_kb=hd, a_skb=M * hd, ldb=da, b_kb=da, b_skb=d * da)
    wb = torch.empty(d, 3 * hd, device=_dev())                # image rows d, k = (p, h, j)
    G.p2_pack([(w[p_, h_], False, wb[:, (p_ * na + h_) * da:(p_ * na + h_ + 1) * da], amax) for p_ in range(3) for h_ in range(na)])
    out2 = torch.empty(M, d, device=_dev())
    G.gemm_p2(g, G.P2Image(wb, amax), out2, M, d, 3 * hd, lda=hd, a_kb=hd, a_skb=M * hd, ldb=3 * hd)
    assert torch.equal(out2, ref2)


def test_layernorm_p2_image_and_gemm():
    from lvt_amd.hip import gemm as G, ew, binding as L
    M, d = 4096, 512
    x, w, b = _rand(M, d, scale=3.0), _rand(d, seed=1) + 1.5, _rand(d, seed=2)
    y0, m0, r0 = ew.layernorm_fwd(x, w, b)
    y, yp, m1, r1 = ew.layernorm_fwd_p2(x, w, b)
    assert torch.equal(y, y0) and torch.equal(m0, m1) and torch.equal(r0, r1)
    bound = float(L.amax_of(y))
    assert bound == float(L.amax_of(y0)) and bound >= float(y.abs().max())
    assert torch.equal(_image_view(yp), _image_ref(y, bound))
    wt = _rand(384, d, seed=3)
    ref, out = torch.empty(M, 384, device=_dev()), torch.empty(M, 384, device=_dev())
    G.gemm(y0, wt, ref, M, 384, d)
    G.gemm_p2(G.P2Image(yp, L.amax_of(y)), _pack(wt), out, M, 384, d)
    assert torch.equal(out, ref)


def test_gemm_p2_output_image():
    """The second output: the P2 image of the result under an a-priori bound (here: 4 x the real max, a loose bound)."""
    from lvt_amd.hip import gemm as G, binding as L
    M, N, K = 2048, 512, 512
    a, w, b = _rand(M, K), _rand(N, K, seed=1, scale=0.2), _rand(N, seed=2)
    ref = torch.empty(M, N, device=_dev())
    G.gemm(a, w, ref, M, N, K, flags=L.EPI_BIAS | L.EPI_RELU, bias=b)
    bound = (ref.abs().max() * 4).reshape(1)
    out, img = torch.empty(M, N, device=_dev()), torch.empty(M, N, device=_dev())
    _, im = G.gemm_p2(a, _pack(w), out, M, N, K, flags=L.EPI_BIAS | L.EPI_RELU, bias=b, out_image=img, out_bound=bound)
    assert torch.equal(out, ref)
    assert torch.equal(_image_view(im.data), _image_ref(ref, float(bound)))


@pytest.mark.parametrize("M", [2048, 16384])
def test_gemm_p2_image_a_many_workgroups_fresh_outputs(M):
    """Six LDS-DMA instructions per wave and k-tile in flight, several workgroups per CU one after the other, 24 workgroups
    sharing every A line: the shape that exposed that `s_waitcnt vmcnt(0)` + barrier does not make another wave's LDS-DMA data
    visible (profiles/r06_lds_dma_visibility.txt).  Outputs start as NaN (a stale correct value must not hide a tile that was
    not computed), every launch is repeated, both the single-launch and the 24-batch form of the q/k/v product."""
    from lvt_amd.hip import gemm as G
    d, na, da = 512, 8, 128
    hd = na * da
    x, w = _rand(M, d), _rand(3 * hd, d, seed=1, scale=0.05)
    ref = torch.empty(M, 3 * hd, device=_dev())
    G.gemm(x, w, ref, M, 3 * hd, d)
    xi, wi = _pack(x), _pack(w)
    kw = dict(lda=d, ldb=d, ldc=hd, batch_outer=3, batch_inner=na, sB=(na * da * d, da * d), sC=(M * hd, da))
    for _ in range(4):
        c1 = torch.full((M, 3 * hd), float("nan"), device=_dev())
        G.gemm_p2(xi, wi, c1, M, 3 * hd, d)
        assert torch.equal(c1, ref)
        c2 = torch.full((3, M, hd), float("nan"), device=_dev())
        G.gemm_p2(xi, wi, c2, M, da, d, **kw)
        assert torch.equal(c2.permute(1, 0, 2).reshape(M, 3 * hd), ref)
        c3 = torch.full((3, M, hd), float("nan"), device=_dev())
        G.gemm_p2(x, wi, c3, M, da, d, **kw)
        assert torch.equal(c3, c2)


def test_dsfvt_loss_with_p2_images_is_bit_identical():
    """The model paths with P2 images -- LVT_P2=qkv (q/k/v weight image, fp32 A) and the full one (vt_attention.P2_IMAGES =
    True: LayerNorm outputs + q/k/v and first-FFN weights as images, both operands of those products by LDS-DMA): the DSFVT
    training loss and every gradient equal the engine-only path's (P2_IMAGES = False) bit for bit, on
    fresh models from the same seed (the shape that caught the LDS-DMA visibility hole: 8 slices, 24 batches per q/k/v launch)."""
    import lvt_amd.modeling.autoregressive.vt_attention as VA
    from lvt_amd.data.dataset_mapper import prepare_slices_batch
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    from util_models import dsfvt_cfg

    def run(p2):
        VA.P2_IMAGES = None if p2 == "qkv" else p2
        if p2 == "qkv":
            os.environ["LVT_P2"] = "qkv"
        try:
            cfg = dsfvt_cfg("cuda:0")
            cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
            torch.manual_seed(13)
            model = build_model(cfg)
            model.train()
            v = cfg.MODEL.AUTOREGRESSIVE.VT
            g = torch.Generator().manual_seed(7)
            codes = torch.randint(0, v.NV, (8, 16, v.NC, 16, 16), generator=g).to("cuda:0")
            abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (8,), generator=g)]
            ctx, sl, sidx, ign = prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)
            with EventStorage(0):
                loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
            loss.backward()
            used = any(getattr(m, "_p2", None) is not None for m in model.modules())
            return float(loss.detach()), {n: p.grad.clone() for n, p in model.model.named_parameters() if p.grad is not None}, used
        finally:
            VA.P2_IMAGES = None
            os.environ.pop("LVT_P2", None)
    l0, g0, u0 = run(False)
    assert not u0
    for mode in (True, "qkv"):     # "full" (both operands of the LayerNorm-fed products), and the q/k/v weight image only
        junk = torch.full((1 << 26,), float("nan"), device="cuda:0")      # dirty the allocator's free blocks between the runs
        del junk
        l1, g1, u1 = run(mode)
        assert u1, mode
        assert l0 == l1
        assert g0.keys() == g1.keys()
        for n in g0:
            assert torch.equal(g0[n], g1[n]), (mode, n)
