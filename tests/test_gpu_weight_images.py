"""Weight tiles as ready LDS images (lvt_conv3d_weight_images + LVT_CONV_WEIGHT_IMAGE, ABI 610): the frame-resident convolutions
stage their weight tiles by LDS-DMA from an image made once per pass instead of splitting the fp32 tile in every workgroup.
Same split function, same bits in LDS: every result must be BIT-IDENTICAL to the in-kernel split (and therefore keeps the parity of
tests/test_gpu_engine.py / test_gpu_vqvae.py against torch, the oracle and the reference's fixtures).  The outputs are allocated in
memory that was filled with NaN right before (a stale correct result in a recycled block must not hide a tile that never landed)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def f16x2():
    from lvt_amd.hip import binding as L
    before = L.get_math_mode()
    L.set_math_mode("f16x2")
    yield
    L.set_math_mode(before)


def _dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1).to(_dev())


def _poison(numel):
    """The next torch.empty of this size gets a block full of NaN."""
    t = torch.full((numel,), float("nan"), device=_dev())
    del t


def _both(fn):
    """fn(images: bool) -> tensor, run without and with the images."""
    from lvt_amd.hip import gemm as G
    out = []
    for images in (False, True):
        G.PackBatch.IMAGES = images
        try:
            out.append(fn(images))
        finally:
            G.PackBatch.IMAGES = True
    return out


@pytest.mark.parametrize("Ci,Co,N", [(256, 256, 3), (128, 256, 70), (256, 128, 5), (32, 128, 2), (256, 256, 512)])
def test_3x3_forward_and_backward_data_bit_identical(Ci, Co, N):
    from lvt_amd.hip import gemm as G, binding as L
    x, w, b = _rand(N, 1, 16, 16, Ci), _rand(Co, Ci, 1, 3, 3, seed=1) * 0.1, _rand(Co, seed=2)
    res = _rand(N, 1, 16, 16, Co, seed=3)
    gy, rx, msrc = _rand(N, 1, 16, 16, Co, seed=5), _rand(N, 1, 16, 16, Ci, seed=6), _rand(N, 1, 16, 16, Ci, seed=7)
    g = G.conv_geom(N, 1, 16, 16, Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))

    def run(images):
        pb = G.PackBatch()
        wp = pb.plain(g, w, Ci, Co)
        wt = pb.t(g, w, Ci, Co) if G.bwd_data_as_conv(g) else None
        pb.launch()
        assert bool(getattr(wp, "_lvt_wimg", False)) == (images and Co % 128 == 0)
        _poison(N * 256 * Co)
        y = G.conv_fwd(g, x, wp, bias=b, res=res, flags=L.EPI_RELU)
        if wt is None:
            return y, None
        assert bool(getattr(wt, "_lvt_wimg", False)) == (images and Ci % 128 == 0)
        _poison(N * 256 * Ci)
        dx = G.conv_bwd_data(g, gy, None, res=rx, mask=msrc, wt=wt)
        return y, dx

    (y0, dx0), (y1, dx1) = _both(run)
    assert torch.isfinite(y1).all() and torch.equal(y0, y1)
    if dx0 is not None:
        assert torch.isfinite(dx1).all() and torch.equal(dx0, dx1)


@pytest.mark.parametrize("Ci,Co,N", [(128, 256, 3), (64, 128, 5), (128, 256, 512)])
def test_stride2_parity_and_phase_forms_bit_identical(Ci, Co, N):
    """Conv2d k4 s2 p1 of 32x32 frames by parity classes, and its transposed pass by output phases."""
    from lvt_amd.hip import gemm as G, binding as L
    x, w, b = _rand(N, 1, 32, 32, Ci), _rand(Co, Ci, 1, 4, 4, seed=1) * 0.1, _rand(Co, seed=2)
    res = _rand(N, 1, 16, 16, Co, seed=4)
    gy, bx = _rand(N, 1, 16, 16, Co, seed=5), _rand(Ci, seed=6)
    rx, msrc = _rand(N, 1, 32, 32, Ci, seed=7), _rand(N, 1, 32, 32, Ci, seed=8)
    g = G.conv_geom(N, 1, 32, 32, Ci, Co, (1, 4, 4), (1, 2, 2), (0, 1, 1))
    assert G.fwd_by_parity(g)
    by_phases = G.bwd_data_by_phases(g)

    def run(images):
        pb = G.PackBatch()
        wp, wq = pb.plain(g, w, Ci, Co), pb.parity(g, w, Ci, Co)
        wph = pb.phases(g, w, Ci, Co) if by_phases else None
        pb.launch()
        _poison(N * 256 * Co)
        y = G.conv_fwd(g, x, wp, bias=b, res=res, flags=L.EPI_RELU, wq=wq)
        dx = None
        if wph is not None:
            _poison(N * 1024 * Ci)
            dx = G.conv_bwd_data(g, gy, wp, bias=bx, res=rx, mask=msrc, wph=wph)
        return y, dx

    (y0, dx0), (y1, dx1) = _both(run)
    assert torch.isfinite(y1).all() and torch.equal(y0, y1)
    if dx0 is not None:
        assert torch.isfinite(dx1).all() and torch.equal(dx0, dx1)


def test_image_bytes_are_the_f16x2_split_of_the_pack():
    """hi + 2^-11 lo of every image element reconstructs w * s to the 22 bits of the arithmetic, pads aside."""
    from lvt_amd.hip import gemm as G, binding as L
    Ci, Co = 64, 256
    w = _rand(Co, Ci, 1, 3, 3, seed=1) * 0.37
    g = G.conv_geom(2, 1, 16, 16, Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    pb = G.PackBatch()
    wp = pb.plain(g, w, Ci, Co)
    base = wp._base
    pb.launch()
    rows, cols = 9 * Ci, Co
    nbytes = L.lib().lvt_conv3d_weight_image_bytes(rows, cols)
    assert nbytes == (rows // 32) * (cols // 128) * 20992 and base.numel() == rows * cols + nbytes // 4
    img = base[rows * cols:].view(torch.float16).view(rows // 32, cols // 128, 2, 5248).float()
    amax = float(L.amax_of(wp))
    import math
    s = 2.0 ** (14 - (math.frexp(amax)[1] - 1))           # lvt_f16_scale: max |w| s in [2^14, 2^15)
    n = torch.arange(128, device=_dev())
    r = ((n & 3) << 5) + (n >> 2)                          # hrow<128>: physical row of logical row n, 40 halfs each + 32 per 32 rows
    off = r * 40 + (r >> 5) * 32
    idx = off[:, None] + torch.arange(32, device=_dev())[None, :]                       # (n, k)
    hi, lo = img[:, :, 0][:, :, idx], img[:, :, 1][:, :, idx]                           # (kt, nt, n, k)
    rec = (hi + lo / 2048.0).permute(0, 3, 1, 2).reshape(rows, cols)                    # (kt, k, nt, n)
    want = wp.view(rows, cols) * s
    assert float((rec - want).abs().max()) <= 2.0 ** -22 * amax * s * 1.01
