"""Residual conv encoder (reference: vidgen/modeling/encoder/resencoder.py:25-76).

Same constructor / `from_config` / state_dict keys (`layers.N.weight`, `layers.N.block.M.weight`);
the forward pass is one fused chain of implicit-GEMM kernels on channels-last activations:

    stride 4:  Conv(k4 s2 p1)+ReLU, Conv(k4 s2 p1)+ReLU, Conv(k3 p1), n x ResBlock

The reference's ResBlock starts with an in-place ReLU, so the block input itself is rectified and
the skip adds relu(x).  That ReLU is therefore folded into the epilogue of the layer
that PRODUCES x, which is what makes the whole stack a chain of conv+epilogue kernels.
"""
from torch import nn

from ...hip.convnet import Layer
from .. import convstack
from .build import ENCODER_REGISTRY
from .encoder import Encoder


@ENCODER_REGISTRY.register()
class ResEncoder(Encoder):
    @classmethod
    def from_config(cls, cfg, **kwargs):
        e = cfg.MODEL.ENCODER
        return cls(in_channels=kwargs.get("in_channels", e.IN_CHANNELS), nf=e.NF, res_channels=e.RES_CHANNELS,
                   norm=e.NORM, use_spectral_norm=e.SPECTRAL, n_layers=e.N_LAYERS,
                   out_activation=e.OUT_ACTIVATION, stride=kwargs.get("stride", 4))

    def __init__(self, in_channels, nf, res_channels, norm, use_spectral_norm, n_layers, out_activation, stride):
        super().__init__()
        convstack.check_norm(norm, use_spectral_norm)
        if out_activation != "":
            raise NotImplementedError("ResEncoder out_activation %r is not used by any shipped config" % out_activation)
        if stride == 4:
            mods = [nn.Conv2d(in_channels, nf // 2, 4, 2, 1), nn.ReLU(True), nn.Conv2d(nf // 2, nf, 4, 2, 1),
                    nn.ReLU(True), nn.Conv2d(nf, nf, 3, 1, 1)]
        elif stride == 2:
            mods = [nn.Conv2d(in_channels, nf // 2, 4, 2, 1), nn.ReLU(True), nn.Conv2d(nf // 2, nf, 3, 1, 1)]
        else:
            raise ValueError
        mods += [convstack.ResBlock(nf, res_channels) for _ in range(n_layers)]
        self.layers = nn.Sequential(*mods)
        self.in_channels, self.out_channels = in_channels, nf
        self._plan = self._build_plan()

    def _build_plan(self):
        """Translate the module list into fused engine layers + the (weight, bias) modules they use."""
        mods = list(self.layers)
        plan, owners = [], []
        for i, m in enumerate(mods):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            # this output is rectified if an explicit ReLU or a ResBlock (in-place ReLU) follows
            relu_after = isinstance(nxt, (nn.ReLU, convstack.ResBlock))
            if isinstance(m, nn.Conv2d):
                k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
                plan.append(Layer("conv", (1, k, k), (1, s, s), (0, p, p), m.in_channels, m.out_channels,
                                  act="relu" if relu_after else ""))
                owners.append(m)
            elif isinstance(m, convstack.ResBlock):
                c3, c1 = m.block[1], m.block[3]
                src = len(plan) - 1          # output index of the (rectified) block input
                plan.append(Layer("conv", (1, 3, 3), (1, 1, 1), (0, 1, 1), c3.in_channels, c3.out_channels, act="relu"))
                owners.append(c3)
                plan.append(Layer("conv", (1, 1, 1), (1, 1, 1), (0, 0, 0), c1.in_channels, c1.out_channels,
                                  act="relu" if relu_after else "", res_from=src))
                owners.append(c1)
        self._owners = owners
        return plan

    def forward_cl(self, x_cl):
        """(N,1,H,W,Cin_pad4) channels-last -> (N,1,H/4,W/4,nf)."""
        return convstack.run_stack(x_cl, self._plan, [(m.weight, m.bias) for m in self._owners])

    def forward(self, x):
        """(N,C,H,W) -> (N,nf,H/4,W/4), the reference's layout contract."""
        y = self.forward_cl(convstack._LayoutIn.apply(x))
        return convstack._LayoutOut.apply(y, self.out_channels)
