"""Residual conv decoder (reference: vidgen/modeling/generator/resdecoder.py:25-75).

    Conv(k3 p1), n x ResBlock, ReLU, ConvT(k4 s2 p1)+ReLU, ConvT(k4 s2 p1), tanh

ConvTranspose layers run as the backward-data form of the implicit-GEMM engine, decomposed by
output stride phase so that no matrix-core work is spent on structurally-zero taps.
"""
from torch import nn

from ...hip.convnet import Layer
from .. import convstack
from .build import GENERATOR_REGISTRY
from .generator import Generator


@GENERATOR_REGISTRY.register()
class ResDecoder(Generator):
    @classmethod
    def from_config(cls, cfg, **kwargs):
        g = cfg.MODEL.GENERATOR
        return cls(in_channels=g.IN_CHANNELS, nf=g.NF, res_channels=g.RES_CHANNELS, out_channels=g.OUT_CHANNELS,
                   norm=g.NORM, use_spectral_norm=g.SPECTRAL, n_layers=g.N_LAYERS,
                   out_activation=kwargs.get("out_activation", g.OUT_ACTIVATION), stride=kwargs.get("stride", 4))

    def __init__(self, in_channels, nf, res_channels, out_channels, norm, use_spectral_norm, n_layers,
                 out_activation, stride):
        super().__init__()
        convstack.check_norm(norm, use_spectral_norm)
        mods = [nn.Conv2d(in_channels, nf, 3, 1, 1)]
        mods += [convstack.ResBlock(nf, res_channels) for _ in range(n_layers)]
        mods.append(nn.ReLU(True))
        if stride == 4:
            mods += [nn.ConvTranspose2d(nf, nf // 2, 4, 2, 1), nn.ReLU(True),
                     nn.ConvTranspose2d(nf // 2, out_channels, 4, 2, 1)]
        elif stride == 2:
            mods += [nn.ConvTranspose2d(nf, out_channels, 4, 2, 1)]
        else:
            raise ValueError
        if out_activation == "tanh":
            mods.append(nn.Tanh())
        elif out_activation != "":
            raise NotImplementedError("ResDecoder out_activation %r is not used by any shipped config" % out_activation)
        self.layers = nn.Sequential(*mods)
        self.in_channels, self.out_channels = in_channels, out_channels
        self._plan = self._build_plan()

    def _build_plan(self):
        mods = list(self.layers)
        plan, owners = [], []
        for i, m in enumerate(mods):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, (nn.ReLU, convstack.ResBlock)):
                act = "relu"
            elif isinstance(nxt, nn.Tanh):
                act = "tanh"
            else:
                act = ""
            if isinstance(m, nn.Conv2d):
                k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
                plan.append(Layer("conv", (1, k, k), (1, s, s), (0, p, p), m.in_channels, m.out_channels, act=act))
                owners.append(m)
            elif isinstance(m, nn.ConvTranspose2d):
                k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
                plan.append(Layer("convT", (1, k, k), (1, s, s), (0, p, p), m.in_channels, m.out_channels, act=act))
                owners.append(m)
            elif isinstance(m, convstack.ResBlock):
                c3, c1 = m.block[1], m.block[3]
                src = len(plan) - 1
                plan.append(Layer("conv", (1, 3, 3), (1, 1, 1), (0, 1, 1), c3.in_channels, c3.out_channels, act="relu"))
                owners.append(c3)
                plan.append(Layer("conv", (1, 1, 1), (1, 1, 1), (0, 0, 0), c1.in_channels, c1.out_channels, act=act,
                                  res_from=src))
                owners.append(c1)
        self._owners = owners
        return plan

    def forward_cl(self, z_cl):
        """(N,1,h,w,Cin) channels-last -> (N,1,4h,4w,Cout_pad4)."""
        return convstack.run_stack(z_cl, self._plan, [(m.weight, m.bias) for m in self._owners])

    def forward(self, z):
        y = self.forward_cl(convstack._LayoutIn.apply(z))
        return convstack._LayoutOut.apply(y, self.out_channels)
