"""The max-|.| plumbing of the f16x2 arithmetic (include/lvt_hip.h: lvt_amax_io, lvt_amax*, the *_amax outputs of the helper
kernels): every reported bound is checked against torch on the device, bit for bit where the kernel reduces and by
inequality + formula where it stores an a-priori bound; stale records are never used; non-finite operands poison the result
instead of vanishing."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def f16x2_mode():
    from lvt_amd.hip import binding as L
    before = L.get_math_mode()
    L.set_math_mode("f16x2")
    yield
    L.set_math_mode(before)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(DEV)


def test_amax_kernels_match_torch():
    from lvt_amd.hip import binding as L
    lib = L.lib()
    for n in (1, 3, 4, 1000, 65537, 1 << 22):
        x = _rand(n, seed=n, scale=3.0)
        x[n // 2] = -7.25                                   # the max sits on a negative entry
        out = torch.zeros(1, device=DEV)
        L.check(lib.lvt_amax(L.ptr(x), n, L.ptr(out), L.stream_ptr()), "lvt_amax")
        assert float(out) == 7.25
        # misaligned view (the scalar path) and an existing larger value (the kernel only ever raises the slot)
        y = torch.cat([torch.zeros(1, device=DEV), x])[1:]
        out = torch.full((1,), 9.5, device=DEV)
        L.check(lib.lvt_amax(L.ptr(y), n, L.ptr(out), L.stream_ptr()), "lvt_amax")
        assert float(out) == 9.5
    # 150 tensors of ragged sizes in one call (three launches of <= 64)
    ts = [_rand(1 + 37 * i, seed=i, scale=0.1 * (1 + i)) for i in range(150)]
    outs = torch.zeros(150, device=DEV)
    arr = (L.AmaxEntry * 150)()
    for i, (e, t) in enumerate(zip(arr, ts)):
        e.x, e.n, e.out = t.data_ptr(), t.numel(), outs[i:i + 1].data_ptr()
    L.check(lib.lvt_amax_multi(arr, 150, L.stream_ptr()), "lvt_amax_multi")
    assert torch.equal(outs, torch.stack([t.abs().max() for t in ts]))
    a, b, o = torch.tensor([2.0], device=DEV), torch.tensor([5.0], device=DEV), torch.tensor([3.0], device=DEV)
    L.check(lib.lvt_amax_merge(L.ptr(a), L.ptr(b), L.ptr(o), L.stream_ptr()), "lvt_amax_merge")
    assert float(o) == 5.0
    L.check(lib.lvt_amax_merge(L.ptr(a), None, L.ptr(o), L.stream_ptr()), "lvt_amax_merge")
    assert float(o) == 5.0


def test_producers_report_the_max_of_what_they_write():
    """Engine epilogues (GEMM incl. ACCUM / PLANES, frame-resident and implicit-GEMM convolutions), LayerNorm backward,
    the layout / loss kernels: the record attached to the output equals torch's abs().max() of it -- no stand-alone pass."""
    from lvt_amd.hip import binding as L, ew, gemm as G
    n0 = L.AMAX_FALLBACKS[0]
    a, b = _rand(300, 512), _rand(200, 512, seed=1)
    L.amax_of(a), L.amax_of(b)
    base = L.AMAX_FALLBACKS[0]
    c = torch.empty(300, 200, device=DEV)
    G.gemm(a, b, c, 300, 200, 512, flags=L.EPI_RELU)
    assert float(L.amax_of(c)) == float(c.abs().max())
    G.gemm(a, b, c, 300, 200, 512, flags=L.EPI_ACCUM)                         # the record follows the accumulated values
    assert float(L.amax_of(c)) == float(c.abs().max())
    planes = torch.empty(3, 300, 200, dtype=torch.bfloat16, device=DEV)
    G.gemm(a, b, planes, 300, 200, 512, flags=L.EPI_PLANES, c_plane=300 * 200)
    assert getattr(planes, "_lvt_amax", None) is None                          # (bf16 images carry no record)
    g = G.conv_geom(3, 1, 16, 16, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x, w = _rand(3, 1, 16, 16, 128, seed=2), _rand(128, 128, 1, 3, 3, seed=3, scale=0.1)
    L.amax_of(x)
    y = G.conv_fwd(g, x, G.pack_weight(g, w, 128, 128), flags=L.EPI_RELU)     # frame-resident kernel
    assert float(L.amax_of(y)) == float(y.abs().max())
    g1 = G.conv_geom(3, 1, 16, 16, 128, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    w1 = _rand(256, 128, 1, 1, 1, seed=4, scale=0.1)
    y1 = G.conv_fwd(g1, x, G.pack_weight(g1, w1, 128, 256))                   # 1x1: the wide GEMM kernel
    assert float(L.amax_of(y1)) == float(y1.abs().max())
    assert rel_err(y1.view(-1, 256), x.view(-1, 128) @ w1.view(256, 128).t()) < 2e-5
    dx1 = G.conv_bwd_data(g1, y1, G.pack_weight(g1, w1, 128, 256))
    assert float(L.amax_of(dx1)) == float(dx1.abs().max())
    assert rel_err(dx1.view(-1, 128), y1.view(-1, 256) @ w1.view(256, 128)) < 2e-5
    # LayerNorm: backward reduces, forward stores the a-priori bound max|w| sqrt(d-1) + max|b| (and it IS a bound)
    t, lw, lb = _rand(1000, 512, seed=5, scale=4.0), _rand(512, seed=6), _rand(512, seed=7)
    yn, mean, rstd = ew.layernorm_fwd(t, lw, lb)
    bound = float(L.amax_of(yn))
    assert abs(bound - (float(lw.abs().max()) * math.sqrt(511.0) + float(lb.abs().max()))) < 1e-5 * bound
    assert bound >= float(yn.abs().max())
    dxl, _, _ = ew.layernorm_bwd(_rand(1000, 512, seed=8), t, mean, rstd, lw, add=_rand(1000, 512, seed=9))
    assert float(L.amax_of(dxl)) == float(dxl.abs().max())
    img = _rand(6, 3, 64 * 64, seed=10)
    cl = ew.to_channels_last(img, 4, 1, torch.full((3,), 0.5, device=DEV), torch.full((3,), 0.25, device=DEV))
    assert float(L.amax_of(cl)) == float(cl.abs().max())
    gm = ew.mse_bwd(cl, _rand(*cl.shape, seed=11), denom=cl.numel(), tanh_of_a=False)
    assert float(L.amax_of(gm)) == float(gm.abs().max())
    gt = ew.tanh_bwd(gm, torch.tanh(cl))
    assert float(L.amax_of(gt)) == float(gt.abs().max())
    # everything above came from records: since `base` only the inputs nothing produced were scanned -- x, the two conv
    # weights (their packs inherit the record), the LayerNorm weight and bias
    assert L.AMAX_FALLBACKS[0] - base == 5, L.AMAX_FALLBACKS[0] - base


def test_operand_spanning_two_tensors_and_nonfinite_operands():
    from lvt_amd.hip import binding as L, gemm as G
    # a batched launch whose second batch lives in another allocation with 1000x larger values: the second bound must count
    dy0, x0 = _rand(2048, 256, seed=1), _rand(2048, 128, seed=2)
    dy1, x1 = _rand(2048, 256, seed=3, scale=1000.0), _rand(2048, 128, seed=4, scale=500.0)
    dw = torch.empty(2, 256, 128, device=DEV)
    es = 4
    G.gemm(dy0, x0, dw, 256, 128, 2048, ta=1, tb=1, lda=256, ldb=128, batch_inner=2,
           sA=(0, (dy1.data_ptr() - dy0.data_ptr()) // es), sB=(0, (x1.data_ptr() - x0.data_ptr()) // es), sC=(0, 256 * 128),
           splits=2, a_also=dy1, b_also=x1)
    assert rel_err(dw[0], dy0.t() @ x0) < 2e-5 and rel_err(dw[1], dy1.t() @ x1) < 2e-5
    # inf / nan in an operand: the result is non-finite where the reference's is (never a silently finite number)
    a, b = _rand(256, 64, seed=5), _rand(128, 64, seed=6)
    a[3, 7] = float("inf")
    b[5, 9] = float("nan")
    c = torch.empty(256, 128, device=DEV)
    G.gemm(a, b, c, 256, 128, 64)
    ref = a @ b.t()
    assert bool((~torch.isfinite(c[3])).all()) and bool((~torch.isfinite(c[:, 5])).all())
    fin = torch.isfinite(ref)
    assert bool(torch.isfinite(c[fin]).all()) and rel_err(c[fin], ref[fin]) < 2e-5


def test_deferred_ema_update_equals_the_blocking_form():
    """DVQEmbedding.straight_through_cl(defer=True) + finish_ema() (the statistics all-reduce runs beside the decoder forward)
    == the one-call form, bit for bit: z_q_st, z_q_bar, indices and the EMA state."""
    import copy
    from lvt_amd.modeling.vq import DVQEmbedding
    torch.manual_seed(3)
    q1 = DVQEmbedding(4, 512, 256, True).to(DEV)
    q2 = copy.deepcopy(q1)
    z = _rand(8, 1, 16, 16, 256, seed=12, scale=0.01)
    st1, bar1 = q1.straight_through_cl(z)
    st2 = q2.straight_through_cl(z, defer=True)
    with pytest.raises(Exception):
        q2.straight_through_cl(z, defer=True)              # the previous pass was never finished
    bar2 = q2.finish_ema()
    assert torch.equal(st1, st2) and torch.equal(bar1, bar2) and torch.equal(q1.last_indices, q2.last_indices)
    for (k, v1), (_, v2) in zip(q1.state_dict().items(), q2.state_dict().items()):
        assert torch.equal(v1, v2), k
    with pytest.raises(Exception):
        q2.finish_ema()                                     # nothing pending


def test_in_place_softmax_drops_the_stale_record():
    """lvt_attn_softmax_fwd rewrites the scores in place through a raw pointer; the max |q k^T| record of the
    buffer must not survive it (with tiny q / k the probabilities, up to 1, would be scaled by a bound ~1e-6 and overflow
    fp16).  The unfused three-launch path on tiny operands under f16x2, with every used record verified (LVT_AMAX_CHECK)."""
    import math
    from lvt_amd.hip import binding as L, gemm as G, tx
    if L.get_math_mode() != "f16x2":
        pytest.skip("f16x2 only")
    B, H, S, da = 1, 8, 256, 128
    hd = H * da
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B * S, hd, generator=g) * sc for sc in (1e-3, 1e-3, 1.0))
    banks = [torch.zeros(H, 2 * n - 1) for n in (1, 16, 16)]
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    bd = [b.to(DEV) for b in banks]
    old, L.AMAX_CHECK = L.AMAX_CHECK, True
    try:
        P = torch.empty(B, H, S, S, device=DEV)
        G.gemm(qd, kd, P, S, S, da, ta=0, tb=0, lda=hd, ldb=hd, ldc=S, batch_outer=B, batch_inner=H,
               sA=(S * hd, da), sB=(S * hd, da), sC=(H * S * S, S * S))
        assert getattr(P, "_lvt_amax", None) is not None
        tx.attn_softmax_fwd_(P, math.sqrt(da), bd[0], bd[1], bd[2], (1, 16, 16), False)
        assert getattr(P, "_lvt_amax", None) is None
        o = torch.empty(B * S, hd, device=DEV)
        G.gemm(P, vd, o, S, da, S, ta=0, tb=1, lda=S, ldb=hd, ldc=hd, batch_outer=B, batch_inner=H,
               sA=(H * S * S, S * S), sB=(S * hd, da), sC=(S * hd, da))
    finally:
        L.AMAX_CHECK = old
    assert torch.isfinite(o).all()
    ref = (torch.softmax((q.view(S, H, da).transpose(0, 1).double() @ k.view(S, H, da).transpose(0, 1).double().transpose(1, 2)) / math.sqrt(da), -1)
           @ v.view(S, H, da).transpose(0, 1).double()).transpose(0, 1).reshape(S, hd)
    assert float((o.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-5
