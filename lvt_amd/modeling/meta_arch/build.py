from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")
META_ARCH_REGISTRY.__doc__ = "Registry of whole models; entries are called as `obj(cfg)`."


def build_model(cfg):
    """`cfg.MODEL.META_ARCHITECTURE` -> nn.Module (vidgen/modeling/meta_arch/build.py:13-19).
    Does not load weights."""
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
