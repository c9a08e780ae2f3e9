"""Config loading: JSON/YAML -> attribute dict (the reference uses easydict.EasyDict, utils/load_config.py:5-24)."""
import json

import yaml


class AttrDict(dict):
    """Recursive attribute-access dict (keys reachable as cfg.a.b and cfg['a']['b'])."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


EasyDict = AttrDict


def get_config_from_json(jsonfile):
    with open(jsonfile, "r") as f:
        return AttrDict(json.load(f))


def get_config_from_yaml(yamlfile):
    with open(yamlfile, "r") as f:
        return AttrDict(yaml.load(f, Loader=yaml.SafeLoader))
