"""One implementation of the four `build_*` factories of the reference (encoder / generator / autoregressive /
meta-architecture): look the class name up in the component's registry, construct it with `from_config`, check the
abstract base and log the parameter count."""
import logging


def component_builder(registry, section, label, base=None, preload=None):
    """Returns `build(cfg, **kwargs)` for one component kind.

    registry : the component's Registry
    section  : name of the cfg.MODEL sub-node holding NAME (e.g. "ENCODER")
    label    : word used in the parameter-count log line
    base     : zero-argument callable returning the abstract base class instances must derive from
    preload  : zero-argument callable importing the modules that register implementations (lazy, avoids cycles)
    """
    log = logging.getLogger("lvt_amd.modeling." + label)

    def build(cfg, **kwargs):
        if preload is not None:
            preload()
        name = getattr(cfg.MODEL, section).NAME
        obj = registry.get(name).from_config(cfg, **kwargs)
        if base is not None and not isinstance(obj, base()):
            raise AssertionError("%s %r does not derive from %s" % (label, name, base().__name__))
        millions = sum(p.numel() for p in obj.parameters()) / 1e6
        log.info("#params in %s: %sM", label, millions)
        return obj

    build.__name__ = "build_" + label
    build.__doc__ = "cfg.MODEL.%s.NAME -> instance built by the class's `from_config(cfg, **kwargs)`." % section
    return build
