"""Python-file configs -> attribute dictionaries (mmcv.Config.fromfile semantics)."""
import os
import pprint
import runpy
import types


class ConfigDict(dict):
    """dict with attribute access, .get and item assignment (the models assign
    ``opt['color_skip_layers'] = ...`` and read ``opt.x`` / ``opt.get('x', d)``)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name) from None

    def __setattr__(self, name, value):
        self[name] = _wrap(value)

    def __setitem__(self, name, value):
        dict.__setitem__(self, name, _wrap(value))

    def copy(self):
        return ConfigDict(dict.copy(self))

    def to_dict(self):
        return _unwrap(self)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        out = ConfigDict()
        for k, x in v.items():
            dict.__setitem__(out, k, _wrap(x))
        return out
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(_unwrap(x) for x in v)
    return v


class Config:
    """``Config.fromfile('cfg.py')``: run the file, keep its non-dunder, non-module globals."""

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError('file "{}" does not exist'.format(filename))
        if filename.endswith(".py"):
            ns = runpy.run_path(filename)
            cfg = {k: v for k, v in ns.items()
                   if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
        elif filename.endswith((".yml", ".yaml")):
            import yaml
            with open(filename) as f:
                cfg = yaml.safe_load(f)
        elif filename.endswith(".json"):
            import json
            with open(filename) as f:
                cfg = json.load(f)
        else:
            raise IOError("Only py/yml/yaml/json type are supported now!")
        return Config(cfg, filename=filename)

    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError("cfg_dict must be a dict, but got {}".format(type(cfg_dict)))
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict))
        object.__setattr__(self, "_filename", filename)
        text = ""
        if filename and os.path.isfile(filename):
            with open(filename) as f:
                text = f.read()
        object.__setattr__(self, "_text", text)

    filename = property(lambda self: self._filename)
    text = property(lambda self: self._text)

    @property
    def pretty_text(self):
        return "\n".join("{} = {}".format(k, pprint.pformat(_unwrap(v))) for k, v in self._cfg_dict.items()) + "\n"

    def dump(self, file=None):
        if file is None:
            return self.pretty_text
        with open(file, "w") as f:
            f.write(self.pretty_text)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def __repr__(self):
        return "Config (path: {}): {}".format(self._filename, dict.__repr__(self._cfg_dict))
