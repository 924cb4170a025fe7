"""Name -> class registry with the decorator / `get` contract the reference re-exports from fvcore
(vidgen/utils/registry.py:2): `@REG.register()` keyed by `__name__`, `REG.get(name)` raising KeyError."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        if name in self._obj_map:
            raise AssertionError("An object named '{}' was already registered in '{}' registry!".format(
                name, self._name))
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        try:
            return self._obj_map[name]
        except KeyError:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())
