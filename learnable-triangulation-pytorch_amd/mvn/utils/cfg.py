"""Config loading with the reference's surface (mvn/utils/cfg.py:5-9 of the reference):
``load_config(path)`` returns an attribute-style dict of the YAML, so experiments/*.yaml load unchanged.
easydict is not a dependency here; ``ConfigDict`` gives the same recursive attribute access."""
import yaml


class ConfigDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def load_config(path):
    with open(path) as fin:
        return ConfigDict(yaml.safe_load(fin))
