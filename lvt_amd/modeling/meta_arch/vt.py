from torch import nn

from .build import META_ARCH_REGISTRY


@META_ARCH_REGISTRY.register()
class VideoTransformerModel(nn.Module):   # placeholder until the transformer path lands
    def __init__(self, cfg):
        raise NotImplementedError
