"""Tiny yacs-compatible CfgNode: attribute dict, `_BASE_` inheritance, literal-eval of strings."""
import ast
import copy
import os
import yaml

BASE_KEY = "_BASE_"


def _decode(v):
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else init_dict
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in init_dict.items():
            if isinstance(v, dict) and not isinstance(v, CfgNode):
                v = type(self)(v)
            elif not isinstance(v, CfgNode):
                v = _decode(v)
            dict.__setitem__(self, k, v)

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError("Attempted to set {} on an immutable CfgNode".format(name))
        self[name] = value

    @staticmethod
    def load_yaml_with_base(filename, allow_unsafe=False):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f)

        def merge_a_into_b(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and k in b:
                    merge_a_into_b(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base = cfg[BASE_KEY]
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            base_cfg = CfgNode.load_yaml_with_base(base, allow_unsafe)
            del cfg[BASE_KEY]
            merge_a_into_b(cfg, base_cfg)
            return base_cfg
        return cfg

    def merge_from_other_cfg(self, other):
        def rec(a, b, path):
            for k, v in a.items():
                if k not in b:
                    raise KeyError("Non-existent config key: {}".format(".".join(path + [k])))
                if isinstance(v, dict):
                    rec(v, b[k], path + [k])
                else:
                    v = _decode(copy.deepcopy(v))
                    old = b[k]
                    if isinstance(old, tuple) and isinstance(v, list):
                        v = tuple(v)
                    elif isinstance(old, list) and isinstance(v, tuple):
                        v = list(v)
                    dict.__setitem__(b, k, v)
        rec(other, self, [])

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            keys = full_key.split(".")
            for sub in keys[:-1]:
                d = d[sub]
            assert keys[-1] in d, full_key
            v = _decode(v)
            old = d[keys[-1]]
            if isinstance(old, tuple) and isinstance(v, list):
                v = tuple(v)
            dict.__setitem__(d, keys[-1], v)

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        self._immutable(True)

    def defrost(self):
        self._immutable(False)

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._immutable(flag)

    def dump(self, **kwargs):
        def to_dict(n):
            if isinstance(n, CfgNode):
                return {k: to_dict(v) for k, v in n.items()}
            if isinstance(n, tuple):
                return list(n)
            return n
        return yaml.safe_dump(to_dict(self), **kwargs)
