"""Minimal stand-in for the `gym` surface the reference touches."""
from . import spaces, utils, envs  # noqa: F401


class Env:
    metadata = {}
    action_space = None
    observation_space = None

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError
