"""Config node + `get_cfg()` with the reference's surface (vidgen/config/config.py:9-106).

The reference subclasses fvcore's CfgNode (yacs).  Neither package exists in this image, so this is
a self-contained implementation of the behaviour the YAML files and callers rely on:
attribute access, `_BASE_` inheritance relative to the including file, python-literal strings
(`KERNEL: (7, 1, 1)`), `merge_from_file`, `merge_from_list` (trailing `KEY VALUE` CLI overrides),
`clone`, `freeze`/`defrost`, `dump`, and type-checked merging against the defaults tree.
"""
import ast
import copy
import os

import yaml

BASE_KEY = "_BASE_"


def _decode(value):
    if isinstance(value, str):
        try:
            return ast.literal_eval(value)
        except (ValueError, SyntaxError):
            return value
    return value


def _coerce(new, old, key):
    """yacs-style type reconciliation between an override and the default it replaces."""
    if old is None or type(new) is type(old):
        return new
    for src, dst in ((list, tuple), (tuple, list)):
        if isinstance(new, src) and isinstance(old, dst):
            return dst(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, str) and not isinstance(new, str):
        # the defaults tree holds "" for fields that YAML files fill with tuples/numbers is not used
        # by the reference; be permissive like yacs' _check_and_coerce only for known casts
        raise ValueError("Type mismatch for key {}: {} vs {}".format(key, type(new), type(old)))
    if isinstance(new, str) and not isinstance(old, str):
        raise ValueError("Type mismatch for key {}: {} vs {}".format(key, type(new), type(old)))
    return new


class CfgNode(dict):
    _IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode._IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode)
                             else (v if isinstance(v, CfgNode) else _decode(v)))

    # attribute access ---------------------------------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        dict.__setitem__(self, name, value)

    def __setitem__(self, name, value):
        if self.is_frozen():
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        dict.__setitem__(self, name, value)

    # (im)mutability -----------------------------------------------------------------------------
    def is_frozen(self):
        return self.__dict__[CfgNode._IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode._IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = type(self)()
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        new.__dict__[CfgNode._IMMUTABLE] = self.is_frozen()
        return new

    # loading / merging ----------------------------------------------------------------------------
    @staticmethod
    def load_yaml_with_base(filename, allow_unsafe=False):
        """Load a YAML file, recursively resolving `_BASE_` relative to the including file."""
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}

        def merge_a_into_b(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and isinstance(b.get(k), dict):
                    merge_a_into_b(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            base_cfg = CfgNode.load_yaml_with_base(base, allow_unsafe=allow_unsafe)
            merge_a_into_b(cfg, base_cfg)
            return base_cfg
        return cfg

    def merge_from_file(self, cfg_filename, allow_unsafe=True):
        loaded = CfgNode(CfgNode.load_yaml_with_base(cfg_filename, allow_unsafe=allow_unsafe))
        loaded_ver = loaded.get("VERSION", None)
        assert loaded_ver is None or loaded_ver <= self.VERSION, \
            "Cannot merge a v{} config into a v{} config.".format(loaded_ver, self.VERSION)
        self.merge_from_other_cfg(loaded)

    def merge_from_other_cfg(self, other):
        if self.is_frozen():
            raise AttributeError("cannot merge into an immutable CfgNode")
        _merge(other, self, [])

    def merge_from_list(self, cfg_list):
        if self.is_frozen():
            raise AttributeError("cannot merge into an immutable CfgNode")
        assert len(cfg_list) % 2 == 0, "Override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                assert p in node, "Non-existent key: {}".format(full_key)
                node = node[p]
            assert parts[-1] in node, "Non-existent key: {}".format(full_key)
            dict.__setitem__(node, parts[-1], _coerce(_decode(v), node[parts[-1]], full_key))

    def dump(self, **kwargs):
        def plain(n):
            if isinstance(n, CfgNode):
                return {k: plain(v) for k, v in n.items()}
            if isinstance(n, tuple):
                return [plain(v) for v in n]
            return n
        return yaml.safe_dump(plain(self), **kwargs)

    def __repr__(self):
        return "{}({})".format(type(self).__name__, dict.__repr__(self))


def _merge(a, b, path):
    for k, v in a.items():
        full = ".".join(path + [k])
        if k not in b:
            raise KeyError("Non-existent config key: {}".format(full))
        if isinstance(v, dict):
            if not isinstance(b[k], CfgNode):
                raise ValueError("Type mismatch for key {}".format(full))
            _merge(v, b[k], path + [k])
        else:
            dict.__setitem__(b, k, _coerce(_decode(copy.deepcopy(v)), b[k], full))


global_cfg = CfgNode()


def get_cfg():
    """A copy of the default config (vidgen/config/config.py:76-85)."""
    from .defaults import _C
    return _C.clone()


def set_global_cfg(cfg):
    global global_cfg
    global_cfg.clear()
    global_cfg.update(cfg)
