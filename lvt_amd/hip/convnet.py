"""Fused conv-stack executor: runs a chain of (transposed) convolutions with bias / ReLU / tanh /
residual epilogues on the implicit-GEMM engine, and its hand-scheduled backward.

Every activation is stored once, post-activation and channels-last.  ReLU backward is folded into
the epilogue of the kernel that produces the upstream gradient (mask = saved post-ReLU output) and
residual gradients are added in the same epilogue, so the backward chain is two or three engine
launches per layer (bwd-data, bwd-weight [+ bias column-sum where the weight-gradient kernel does not
produce it]) with no stand-alone elementwise pass.

Routing (round 2): the 3x3 and 4x4/stride-2 layers between 16x16 and 32x32 frames run on the
frame-resident kernels -- forward through `conv_fwd` (3x3) / `conv_fwd(wq=...)` (strided, parity classes),
transposed passes through `conv_bwd_data(wt=...)` (3x3, as a forward convolution over transposed weights) /
`conv_bwd_data(wph=...)` (stride 2, phase by phase), weight gradients inside `conv_bwd_weight`.  The extra
weight packs are made next to the forward one; every other geometry takes the implicit-GEMM engine.
"""
import os

import torch

from . import binding as L
from . import gemm as G
from . import ew

BIAS_OF_X = not os.environ.get("LVT_NO_BIAS_OF_X")      # A/B switch: transposed layers' bias gradient inside the weight-gradient launch


class Layer:
    """One conv / ConvTranspose layer of a stack.

    kind: "conv" | "convT".  For convT the geometry is that of the equivalent forward conv whose
    backward-data IS this layer (Ci = out channels of the ConvTranspose, Co = its in channels).
    act: "" | "relu" | "tanh".  res_from: index of an earlier output added before the activation
    (-1: none).  ci_real / co_real: channel counts of the torch-layout weight (pads excluded).
    """

    def __init__(self, kind, kernel, stride, pad, cin, cout, act="", res_from=-1):
        self.kind, self.kernel, self.stride, self.pad = kind, kernel, stride, pad
        self.cin, self.cout, self.act, self.res_from = cin, cout, act, res_from

    @staticmethod
    def pad4(c):
        return (c + 3) // 4 * 4


def _geom(layer, x_shape):
    """Geometry of the forward conv the engine sees, given the channels-last input shape."""
    N, T, H, W, _ = x_shape
    ci, co = Layer.pad4(layer.cin), Layer.pad4(layer.cout)
    if layer.kind == "conv":
        return G.conv_geom(N, T, H, W, ci, co, layer.kernel, layer.stride, layer.pad)
    # ConvTranspose: output extent = (in - 1) * s - 2p + k ; equivalent conv maps output -> input
    To, Ho, Wo = [(i - 1) * s - 2 * p + k for i, s, p, k in zip((T, H, W), layer.stride, layer.pad, layer.kernel)]
    g = G.conv_geom(N, To, Ho, Wo, co, ci, layer.kernel, layer.stride, layer.pad)
    assert (g.To, g.Ho, g.Wo) == (T, H, W), "ConvTranspose geometry mismatch"
    return g


def _padded_bias(layer, b):
    co = Layer.pad4(layer.cout)
    if b is None or b.numel() == co:
        return b
    return torch.cat([b, b.new_zeros(co - b.numel())])


def _act_flag(act):
    return {"": 0, "relu": L.EPI_RELU, "tanh": L.EPI_TANH}[act]


def stack_forward(layers, x, params, want_grad=True):
    """x: (N,T,H,W,C) channels-last.  params: [(weight, bias)] in torch layout.  want_grad: a backward pass will follow
    (the transposed weight packs it needs are made here, next to the forward ones).
    Returns (outs, saved) where outs[i] is the post-activation output of layer i."""
    outs, geoms, packed = [], [], []
    # every weight pack of the stack in ONE launch up front (24 ~5 us launches per VQ-VAE pass otherwise, each in front of the
    # layer that needs it): the geometries follow from the shapes alone
    pb = G.PackBatch()
    shape = tuple(x.shape)
    for i, (ly, (w, b)) in enumerate(zip(layers, params)):
        g = _geom(ly, shape)
        wt = wph = wq = None
        if ly.kind == "conv":
            wp = pb.plain(g, w, ly.cin, ly.cout)
            if want_grad and i > 0 and G.bwd_data_by_phases(g):
                wph = pb.phases(g, w, ly.cin, ly.cout)          # for this layer's backward-data
            if G.fwd_by_parity(g):
                wq = pb.parity(g, w, ly.cin, ly.cout)           # this layer's forward (4x4 / stride 2)
            # backward-data of the 3x3 layers runs as a forward convolution over transposed weights (frame-resident kernel)
            if want_grad and G.bwd_data_as_conv(g):
                wt = pb.t(g, w, ly.cin, ly.cout)
            shape = (g.N, g.To, g.Ho, g.Wo, g.Co)
        else:
            wp = pb.plain(g, w, ly.cout, ly.cin)   # ConvTranspose weight is (in, out, k..) == conv (Co, Ci)
            if G.bwd_data_by_phases(g):
                wph = pb.phases(g, w, ly.cout, ly.cin)          # this layer's FORWARD is a transposed pass
            if want_grad and G.fwd_by_parity(g):
                wq = pb.parity(g, w, ly.cout, ly.cin)           # its backward-data is the strided convolution
            shape = (g.N, g.Ti, g.Hi, g.Wi, g.Ci)
        geoms.append(g)
        packed.append((wp, wt, wph, wq))
    pb.launch()
    cur = x
    for i, (ly, (w, b)) in enumerate(zip(layers, params)):
        g, (wp, wt, wph, wq) = geoms[i], packed[i]
        seen = (g.N, g.Ti, g.Hi, g.Wi) if ly.kind == "conv" else (g.N, g.To, g.Ho, g.Wo)
        assert seen == tuple(cur.shape[:4]), "geometry of the pre-pass does not match the activation"
        res = outs[ly.res_from] if ly.res_from >= 0 else None
        bias = _padded_bias(ly, b)
        if ly.kind == "conv":
            y = G.conv_fwd(g, cur, wp, bias=bias, res=res, flags=_act_flag(ly.act), wq=wq)
        elif (ly.cout <= 3 and ly.kernel == (1, 4, 4) and ly.stride == (1, 2, 2) and ly.pad == (0, 1, 1)
              and ly.cin % 16 == 0 and res is None and ly.act in ("", "tanh")):
            y = G.convT4_fwd(cur, w, b, ly.act == "tanh")          # image-side layer: dedicated kernel
        else:
            y = G.conv_bwd_data(g, cur, wp, bias=bias, res=res, flags=_act_flag(ly.act), wph=wph)
        outs.append(y)
        cur = y
    return outs, (geoms, packed)


def stack_backward(layers, x, outs, saved, grad_out, need_input_grad=False):
    """grad_out: dL/d(outs[-1]) (post-activation).  Returns (grad_x or None, [(dw, db)])."""
    geoms, packed = saved
    n = len(layers)
    # g_pre of the last layer
    last = layers[-1]
    if last.act == "tanh":
        gpre = ew.tanh_bwd(grad_out, outs[-1])
    elif last.act == "relu":
        raise L.LvtError("a ReLU-terminated stack is not used by the reference architectures")
    else:
        gpre = grad_out
    # residual consumers: res_grad[j] = g_pre of the layer that used outs[j] as its residual
    res_user = {ly.res_from: i for i, ly in enumerate(layers) if ly.res_from >= 0}
    gpres = [None] * n
    gpres[n - 1] = gpre
    grads = [None] * n
    for i in range(n - 1, -1, -1):
        ly, g, (wp, wt, wph, wq) = layers[i], geoms[i], packed[i]
        inp = outs[i - 1] if i > 0 else x
        gp = gpres[i]
        # parameter gradients
        co = Layer.pad4(ly.cout)
        db = None
        if ly.kind == "conv":
            dw, db = G.conv_bwd_weight(g, inp, gp, ly.cin, ly.cout, want_bias=True)     # db rides on the dy stream
        else:
            # transposed layer: the same call with the operands swapped; its bias gradient is the column sum of gp, which the
            # stride-2 frame-resident kernel adds up from the patches it stages
            dw, db = G.conv_bwd_weight(g, gp, inp, ly.cout, ly.cin, want_bias=True, bias_of_x=BIAS_OF_X)
            if not BIAS_OF_X:
                db = None
        if db is None:
            db = G.colsum(gp, gp.numel() // co, co)[:ly.cout]
        grads[i] = (dw, db)      # dw is (Co, Ci, Kt, Kh, Kw); callers view it as the parameter shape
        # gradient w.r.t. the layer input == g_pre of layer i-1 (mask / residual folded in)
        if i == 0 and not need_input_grad:
            break
        prev = layers[i - 1] if i > 0 else None
        res = gpres[res_user[i - 1]] if (i - 1) in res_user else None
        mask = outs[i - 1] if (prev is not None and prev.act == "relu") else None
        if prev is not None and prev.act == "tanh":
            raise L.LvtError("tanh is only supported on the last layer of a stack")
        if ly.kind == "conv":
            gin = G.conv_bwd_data(g, gp, wp, res=res, mask=mask, wt=wt, wph=wph)
        else:
            gin = G.conv_fwd(g, gp, wp, res=res, mask=mask, wq=wq)
        if i > 0:
            gpres[i - 1] = gin
        else:
            return gin, grads
    return None, grads
