"""Config registry — the reference's plug-in point (utils/utils.py:27-42): a ``target:`` dotted path + ``params``.

``configs/inference_geo4d.yaml`` in this repo is the reference yaml with only the ``target:`` strings changed, so
``instantiate_from_config(cfg.model)`` builds the MI355X engine instead of the PyTorch modules. omegaconf is not in the
image; ``load_config`` returns attribute-accessible dicts that behave like the OmegaConf nodes the scripts use.
"""
import importlib

import yaml


class Config(dict):
    """dict with attribute access and ``pop(key, default)`` (the subset of OmegaConf the entry scripts rely on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def to_config(x):
    if isinstance(x, dict):
        return Config({k: to_config(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [to_config(v) for v in x]
    return x


def load_config(path):
    with open(path) as f:
        return to_config(yaml.safe_load(f))


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__" or config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))
