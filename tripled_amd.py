"""Alias loader: ``import tripled_amd`` gives the package that lives in the (non-identifier)
directory ``tripled-exploring-depth-estimation-with-self-supervised-representation-learning_amd/``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "tripled-exploring-depth-estimation-with-self-supervised-representation-learning_amd")


def _load():
    name = "tripled_amd"
    spec = importlib.util.spec_from_file_location(name, os.path.join(_DIR, "__init__.py"),
                                                  submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_load()
