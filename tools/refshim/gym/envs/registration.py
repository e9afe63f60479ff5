"""Dict registry standing in for gym.envs.registration (fixture generation only)."""
import importlib

_REGISTRY = {}


def register(id, entry_point=None, **kwargs):
    _REGISTRY[id] = (entry_point, kwargs)


def make(env_id, **kwargs):
    entry_point, reg_kwargs = _REGISTRY[env_id]
    mod_name, cls_name = entry_point.split(":")
    cls = getattr(importlib.import_module(mod_name), cls_name)
    return cls(**dict(reg_kwargs.get("kwargs", {}), **kwargs))
