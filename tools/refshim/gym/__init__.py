"""Minimal stand-in for gym 0.14 so the reference package imports in the fixture container.

Test infrastructure only (used by tools/gen_golden.py); never shipped to the product path.
Only the names the reference touches at import/construct time are provided.
"""
from . import spaces  # noqa: F401
from .envs import registration as _registration


class Env(object):
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)


class _Logger(object):
    @staticmethod
    def set_level(level):
        return None


logger = _Logger()


def make(env_id, **kwargs):
    return _registration.make(env_id, **kwargs)
