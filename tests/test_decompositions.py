"""The index algebra of the frame-resident kernels (csrc/gemm_engine.hip: lvt_conv_patch_kernel<0,1,2>, csrc/conv_wgrad.hip),
restated with torch on the CPU and checked against torch's own convolutions.  The kernels themselves are checked on the GPU
(tests/test_gpu_engine.py); these host tests pin the decompositions and weight-pack formulas they implement, so that a
change of a formula shows up in the CPU suite too.

  * 3x3 / pad 1 over an 18x18 zero-haloed patch: tap (dy, dx) reads patch[y + dy][x + dx]
  * backward-data of a stride-1 conv == forward conv over wt[taps-1-tap][co][ci] = w[co][ci][tap] (pack_weight_t)
  * ConvTranspose k4 s2 p1 by output phases: phase (py, px), tap (a, b) reads patch[y + py + a][x + px + b] with
    w[..][3 - py - 2a][3 - px - 2b]                                   (pack_weight_phases)
  * Conv k4 s2 p1 by parity classes: class (py, px), tap (a, b) reads sub[y + a][x + b], sub[r][c] = in[2r + py - 1][2c + px - 1],
    with w[..][2a + py][2b + px]                                      (pack_weight_parity)
  * weight gradient with swapped roles: acc[t'][co][ci] over the dy patch == dW[8 - t']
  * patch_orow: the GEMM row order of a 256-pixel frame tile
"""
import torch
import torch.nn.functional as F


def _rand(*shape, seed=0):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed + sum(shape)), dtype=torch.float64) * 2 - 1


def _patch(x):                       # (C, 16, 16) -> (C, 18, 18) with zero halo
    return F.pad(x, (1, 1, 1, 1))


def test_conv3x3_over_the_patch_and_backward_data_as_forward_conv():
    ci, co = 5, 7
    x, w = _rand(ci, 16, 16), _rand(co, ci, 3, 3, seed=1)
    p = _patch(x)
    y = torch.zeros(co, 16, 16, dtype=torch.float64)
    for dy in range(3):
        for dx in range(3):
            y += torch.einsum("oc,chw->ohw", w[:, :, dy, dx], p[:, dy:dy + 16, dx:dx + 16])
    assert torch.allclose(y, F.conv2d(x[None], w, padding=1)[0], atol=1e-12)
    # dx of that conv for a gradient g: forward conv of g over wt[tap'] (co -> ci) with tap' = 8 - tap
    g = _rand(co, 16, 16, seed=2)
    wt = torch.stack([w[:, :, (8 - t) // 3, (8 - t) % 3] for t in range(9)])          # (9, co, ci): wt[8 - tap] = w[.., tap]
    gp, dx_ = _patch(g), torch.zeros(ci, 16, 16, dtype=torch.float64)
    for t in range(9):
        dx_ += torch.einsum("oc,ohw->chw", wt[t], gp[:, t // 3:t // 3 + 16, t % 3:t % 3 + 16])
    ref = F.conv_transpose2d(g[None], w, padding=1)[0]
    assert torch.allclose(dx_, ref, atol=1e-12)


def test_transposed_conv_k4s2_by_output_phases():
    cin, cout = 6, 4                                  # ConvTranspose2d(cin -> cout): weight (cin, cout, 4, 4)
    x, w = _rand(cin, 16, 16), _rand(cin, cout, 4, 4, seed=1)
    ref = F.conv_transpose2d(x[None], w, stride=2, padding=1)[0]          # (cout, 32, 32)
    p = _patch(x)
    out = torch.zeros(cout, 32, 32, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    ky, kx = 3 - py - 2 * a, 3 - px - 2 * b
                    out[:, py::2, px::2] += torch.einsum("io,ihw->ohw", w[:, :, ky, kx], p[:, py + a:py + a + 16, px + b:px + b + 16])
    assert torch.allclose(out, ref, atol=1e-12)


def test_strided_conv_k4s2_by_parity_classes():
    ci, co = 5, 3
    x, w = _rand(ci, 32, 32), _rand(co, ci, 4, 4, seed=1)
    ref = F.conv2d(x[None], w, stride=2, padding=1)[0]                    # (co, 16, 16)
    xp = F.pad(x, (1, 2, 1, 2))                                           # xp[i + 1] = x[i]; rows -1 .. 33
    out = torch.zeros(co, 16, 16, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            sub = xp[:, py::2, px::2][:, :17, :17]                        # sub[r][c] = x[2r + py - 1][2c + px - 1]
            for a in range(2):
                for b in range(2):
                    out += torch.einsum("oc,chw->ohw", w[:, :, 2 * a + py, 2 * b + px], sub[:, a:a + 16, b:b + 16])
    assert torch.allclose(out, ref, atol=1e-12)


def test_weight_gradient_with_swapped_roles_reverses_the_taps():
    ci, co = 4, 3
    x, g = _rand(ci, 16, 16), _rand(co, 16, 16, seed=1)
    w = _rand(co, ci, 3, 3, seed=2).requires_grad_(True)
    (F.conv2d(x[None], w, padding=1)[0] * g).sum().backward()
    xp, gp = _patch(x), _patch(g)
    plain = torch.stack([torch.einsum("chw,ohw->oc", xp[:, t // 3:t // 3 + 16, t % 3:t % 3 + 16], g) for t in range(9)])
    swapped = torch.stack([torch.einsum("ohw,chw->oc", gp[:, t // 3:t // 3 + 16, t % 3:t % 3 + 16], x) for t in range(9)])
    for t in range(9):
        assert torch.allclose(plain[t], w.grad[:, :, t // 3, t % 3], atol=1e-12)
        assert torch.allclose(swapped[8 - t], w.grad[:, :, t // 3, t % 3], atol=1e-12)


def test_patch_row_order_is_a_bijection_with_row_contiguous_lane_groups():
    def orow(row):
        f, t32, q, e = row & 255, (row & 255) >> 5, (row >> 2) & 7, row & 3
        y, x = 2 * t32 + (bin(q).count("1") & 1), 4 * (q >> 1) + e
        return (row & ~255) + y * 16 + x
    assert sorted(orow(r) for r in range(512)) == list(range(512))
    # the 16-lane groups a ds_read_b128 is serviced in each cover 16 consecutive pixels of ONE image row
    for group in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                  [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
        for tile in range(8):
            px = sorted(orow(32 * tile + l) for l in group)
            assert px == list(range(px[0], px[0] + 16)) and px[0] % 16 == 0
