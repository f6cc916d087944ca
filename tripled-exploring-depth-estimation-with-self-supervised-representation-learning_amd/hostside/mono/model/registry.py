"""Name -> class registry (reference: mono/model/registry.py:8-42).
``MONO.module_dict[cfg.model['name']](cfg.model)`` is how train.py builds a model (train.py:98-99)."""
import torch.nn as nn


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._module_dict)

    def register_module(self, cls):
        if not (isinstance(cls, type) and issubclass(cls, nn.Module)):
            raise TypeError("module must be a child of nn.Module, but got {}".format(cls))
        if cls.__name__ in self._module_dict:
            raise KeyError("{} is already registered in {}".format(cls.__name__, self._name))
        self._module_dict[cls.__name__] = cls
        return cls

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return "Registry(%s: %s)" % (self._name, sorted(self._module_dict))


MONO = Registry("mono")
SEGMENTATION = Registry("segmentation")
