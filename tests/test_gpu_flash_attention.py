"""The flash attention kernels (csrc/attention_flash.hip: lvt_attn_fwd_flash / lvt_attn_bwd_flash) against an fp64 evaluation
of the reference formula (vidgen/modeling/autoregressive/vt_attention.py:59-81 with the bias of :169-174) and its autograd
backward: output, row statistics, dq / dk / dv and the three bias-bank gradients, both block geometries, masked and not;
the accuracy envelope of the per-row f16x2 split against a plain fp32 evaluation; and the layer through the flash path against
the layer through the plane kernels."""
import math

import pytest
import torch

from conftest import rel_err
from oracle import lvt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S, DA = 256, 128


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * 2 - 1


def _reference(q, k, v, go, banks, blk, masked, dtype=torch.float64):
    """-> o, m, linv, dq, dk, dv, dbanks in `dtype` on the CPU (autograd of the reference formula)."""
    B = q.shape[0] // S
    H = q.shape[1] // DA
    leaves = [t.to(dtype).clone().requires_grad_(True) for t in (q, k, v)]
    bl = [t.to(dtype).clone().requires_grad_(True) for t in banks]
    qh, kh, vh = [t.view(B, S, H, DA).permute(0, 2, 1, 3) for t in leaves]
    bias = O.rel_position_bias(*bl, blk).transpose(0, 1)                                # (1, H, S, S)
    sc = qh @ kh.transpose(2, 3) / math.sqrt(DA) + bias
    if masked:
        sc = sc.masked_fill(torch.triu(torch.ones(S, S), 1).bool(), -1e4)
    P = torch.softmax(sc, -1)
    o = (P @ vh).permute(0, 2, 1, 3).reshape(B * S, H * DA)
    o.backward(go.to(dtype))
    m = sc.max(-1).values * math.log2(math.e)          # the kernels keep the row max in log2 units
    linv = 1.0 / torch.exp(sc - sc.max(-1, keepdim=True).values).sum(-1)
    return [o.detach(), m.detach().reshape(-1), linv.detach().reshape(-1)] + [t.grad for t in leaves] + [t.grad for t in bl]


@pytest.fixture(params=[True, False], ids=["onepass", "twopass"])
def onepass(request):
    """Backward A with delta_i = dO_i . O_i in one pass over the keys (the product path: `o` handed to lvt_attn_bwd_flash), or
    with delta from a first pass over the keys (o == NULL)."""
    return request.param


def _flash(q, k, v, go, banks, blk, masked, onepass=True):
    from lvt_amd.hip import binding as L, tx
    assert L.get_math_mode() == "f16x2"
    B = q.shape[0] // S
    H = q.shape[1] // DA
    qkv = torch.stack([q, k, v]).to(DEV).contiguous()
    bd = [t.to(DEV).contiguous() for t in banks]
    o, stats = tx.attn_fwd_flash(qkv, B, H, S, DA, math.sqrt(DA), bd[0], bd[1], bd[2], blk, masked)
    dqkv, ddt, ddh, ddw = tx.attn_bwd_flash(qkv, go.to(DEV).contiguous(), stats, B, H, S, DA, math.sqrt(DA), bd[0], bd[1], bd[2],
                                            blk, masked, o=o if onepass else None)
    torch.cuda.synchronize()
    return [o, stats[0], stats[1], dqkv[0], dqkv[1], dqkv[2], ddt, ddh, ddw]


NAMES = ["o", "m", "linv", "dq", "dk", "dv", "ddt", "ddh", "ddw"]


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("blk", [(1, 16, 16), (4, 8, 8)])
def test_flash_attention_vs_fp64(masked, blk, onepass):
    B, H = 2, 8
    hd = H * DA
    q, k, v, go = (_rand(B * S, hd, seed=s) for s in (1, 2, 3, 4))
    q = q * 3.0                                                                        # scores of a few units: a peaked softmax
    banks = [_rand(H, 2 * n - 1, seed=5 + i) * 0.5 for i, n in enumerate(blk)]
    ref = _reference(q, k, v, go, banks, blk, masked)
    ref32 = _reference(q, k, v, go, banks, blk, masked, dtype=torch.float32)
    got = _flash(q, k, v, go, banks, blk, masked, onepass)
    bank_scale = max(float(ref[i].abs().max()) for i in (6, 7, 8))
    for n, a, r, r32 in zip(NAMES, got, ref, ref32):
        # (the gradient of a one-entry bank is sum_ij g_ij == 0 up to rounding: judged against the scale of the other banks)
        scale = bank_scale if n.startswith("dd") else float(r.abs().max())
        err = float((a.double().cpu() - r).abs().max())
        err32 = float((r32.double() - r).abs().max())
        assert err < max(2e-5 * scale, 4 * err32), (n, err, err32, scale)
    if masked:       # structural zeros of the causal layer: key 255 receives a gradient from query 255 only
        assert torch.isfinite(got[4]).all() and torch.isfinite(got[5]).all()


@pytest.mark.parametrize("kind", ["row_ladder", "heavy_tail", "tiny", "huge"])
def test_flash_attention_accuracy_envelope(kind, onepass):
    """The per-row split keeps 22 bits relative to each ROW's max: rows of very different magnitude (a ladder down to 2^-20),
    heavy-tailed rows, tiny (1e-6) and huge (1e4) operands must come out no worse than a plain fp32 evaluation of the same
    formula (judged against fp64: error <= 1.5 x the fp32 evaluation's, or 2e-6 of the tensor's max)."""
    B, H, blk, masked = 1, 8, (1, 16, 16), False
    hd = H * DA
    g = torch.Generator().manual_seed(11)
    q, k, v, go = (torch.randn(B * S, hd, generator=g) for _ in range(4))
    if kind == "row_ladder":
        lad = torch.exp2(-20.0 * torch.arange(B * S).float() / (B * S - 1)).view(-1, 1)
        k, v, go = k * lad, v * lad.flip(0), go * lad
    elif kind == "heavy_tail":
        q, k, v, go = (t * torch.exp(1.5 * torch.randn(t.shape, generator=g)) for t in (q, k, v, go))
        q = q * 0.05                                                                    # keep the softmax off the one-hot regime
    elif kind == "tiny":
        q, k, v, go = q * 1e-3, k * 1e-3, v * 1e-6, go * 1e-6
    else:
        q, v, go = q * 0.5, v * 1e4, go * 1e4
    banks = [_rand(H, 2 * n - 1, seed=5 + i) * 0.5 for i, n in enumerate(blk)]
    ref = _reference(q, k, v, go, banks, blk, masked)
    ref32 = _reference(q, k, v, go, banks, blk, masked, dtype=torch.float32)
    got = _flash(q, k, v, go, banks, blk, masked, onepass)
    for n, a, r, r32 in zip(NAMES, got, ref, ref32):
        if n.startswith("dd") or n in ("m", "linv"):
            continue
        scale = float(r.abs().max())
        err = float((a.double().cpu() - r).abs().max())
        err32 = float((r32.double() - r).abs().max())
        assert err <= max(1.5 * err32, 2e-6 * scale), (kind, n, err, err32, scale)


@pytest.mark.parametrize("blk", [(1, 16, 16), (4, 8, 8)])
def test_flash_attention_zero_rows_and_zero_gradients(blk, onepass):
    """Operands with all-zero rows, a (sample, head) whose dO is entirely zero (a block that does not reach the loss: the DSSVT
    encoder has them) and one whose V is zero: every scale derived from a row maximum stays finite -- no inf * 0."""
    B, H = 2, 8
    hd = H * DA
    q, k, v, go = (_rand(B * S, hd, seed=s) for s in (11, 12, 13, 14))
    go[:S] = 0                                  # sample 0: no gradient at all
    go[S:, :DA] = 0                             # sample 1, head 0 alike
    v[S:, DA:2 * DA] = 0                        # sample 1, head 1: V == 0
    q[5] = 0; k[7] = 0; k[S + 9] = 0
    banks = [_rand(H, 2 * n - 1, seed=5 + i) * 0.5 for i, n in enumerate(blk)]
    for masked in (False, True):
        ref = _reference(q, k, v, go, banks, blk, masked)
        got = _flash(q, k, v, go, banks, blk, masked, onepass)
        bank_scale = max(float(ref[i].abs().max()) for i in (6, 7, 8))
        for n, a, r in zip(NAMES, got, ref):
            assert torch.isfinite(a).all(), n
            scale = bank_scale if n.startswith("dd") else float(r.abs().max())
            assert float((a.double().cpu() - r).abs().max()) < 2e-5 * scale, (n, masked)
        assert float(got[3][:S].abs().max()) == 0.0 and float(got[5][:S].abs().max()) == 0.0     # sample 0: dq == dv == 0


@pytest.mark.parametrize("block,masked", [((1, 16, 16), False), ((1, 16, 16), True), ((4, 8, 8), True), ((4, 8, 8), False)])
def test_flash_layer_equals_plane_layer(block, masked):
    """One BlockLocalAttention layer through the flash kernels against the same layer through the plane kernels of
    csrc/attention_pipe.hip (which keep the attention matrix): output and every gradient."""
    import lvt_amd.modeling.autoregressive.vt_attention as A
    torch.manual_seed(0)
    layer = A.BlockLocalAttention(block, 128, 512, 8, masked=masked).to(DEV)
    with torch.no_grad():
        layer.dt_bank.normal_(0, 0.3); layer.dh_bank.normal_(0, 0.3); layer.dw_bank.normal_(0, 0.3)
    x = torch.randn(8 * 256, 512, device=DEV)
    gy = torch.randn_like(x)

    def run(flash):
        A.FLASH_ATTENTION = flash
        try:
            for p in layer.parameters():
                p.grad = None
            xx = x.clone().requires_grad_(True)
            y = layer.forward_tokens(xx, layer.block_size)
            y.backward(gy)
            return [y.detach(), xx.grad] + [p.grad.clone() for p in layer.parameters()]
        finally:
            A.FLASH_ATTENTION = None

    new, old = run(True), run(False)
    names = ["y", "dx"] + [n for n, _ in layer.named_parameters()]
    bank_scale = float(old[names.index("dh_bank")].abs().max())
    for n, a, c in zip(names, new, old):
        scale = bank_scale if n.endswith("_bank") else float(c.abs().max())
        assert float((a - c).abs().max()) < 2e-5 * scale + 1e-30, (n, float((a - c).abs().max()), scale)
