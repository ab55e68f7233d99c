"""Name -> class registries (mirror of models/registers.py:6-8 and
net_utils/registry.py of the reference)."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return f"Registry(name={self._name}, items={list(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key, alter_key=None):
        if key in self._module_dict:
            return self._module_dict[key]
        return self._module_dict.get(alter_key, None)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        if cls.__name__ in self._module_dict:
            raise KeyError(f"{cls.__name__} is already registered in {self._name}")
        self._module_dict[cls.__name__] = cls
        return cls


METHODS = Registry('method')
MODULES = Registry('module')
LOSSES = Registry('loss')
